/* See scg_oracle.h.  Scalar float64 code per env; `#pragma omp parallel for` over envs. */
#include "scg_oracle.h"
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int nx_of(int s) { return s == 1 ? 2 : s == 2 ? 6 : s == 3 ? 12 : 4; }
static int nu_of(int s) { return s == 1 ? 1 : s == 2 ? 2 : s == 3 ? 4 : 1; }

/* quadrotor.py:594-676 */
static void f_eval(const scgo_config* c, const double* x, const double* u, double* d) {
    if (c->system == 1) {
        d[0] = x[1];
        d[1] = u[0] / c->m - c->g;
    } else if (c->system == 2) {
        double T = u[0] + u[1];
        d[0] = x[1];
        d[1] = sin(x[4]) * T / c->m;
        d[2] = x[3];
        d[3] = cos(x[4]) * T / c->m - c->g;
        d[4] = x[5];
        d[5] = c->l * (u[1] - u[0]) / c->iyy / sqrt(2.0);
    } else if (c->system == 3) {
        double phi = x[6], th = x[7], psi = x[8], p = x[9], q = x[10], r = x[11];
        double sph = sin(phi), cph = cos(phi), sth = sin(th), cth = cos(th), sps = sin(psi), cps = cos(psi);
        double T = u[0] + u[1] + u[2] + u[3];
        double s2 = sqrt(2.0);
        double Mx = c->l / s2 * (u[0] + u[1] - u[2] - u[3]);
        double My = c->l / s2 * (-u[0] + u[1] + u[2] - u[3]);
        double Mz = c->km / c->kf * (u[0] - u[1] + u[2] - u[3]);
        double tth = tan(th);
        d[0] = x[1];
        d[1] = (cps * sth * cph + sps * sph) * T / c->m;
        d[2] = x[3];
        d[3] = (sps * sth * cph - cps * sph) * T / c->m;
        d[4] = x[5];
        d[5] = cth * cph * T / c->m - c->g;
        d[6] = p + sph * tth * q + cph * tth * r;
        d[7] = cph * q - sph * r;
        d[8] = sph / cth * q + cph / cth * r;
        d[9] = (Mx - (q * c->izz * r - r * c->iyy * q)) / c->ixx;
        d[10] = (My - (r * c->ixx * p - p * c->izz * r)) / c->iyy;
        d[11] = (Mz - (p * c->iyy * q - q * c->ixx * p)) / c->izz;
    } else {
        double th = x[2], thd = x[3];
        double Mm = c->cp_m + c->cp_M, ml = c->cp_m * c->cp_l;
        double s = sin(th), co = cos(th);
        double tmp = (u[0] + ml * thd * thd * s) / Mm;
        double thdd = (c->g * s - co * tmp) / (c->cp_l * (4.0 / 3.0 - c->cp_m * co * co / Mm));
        d[0] = x[1];
        d[1] = tmp - ml * thdd * co / Mm;
        d[2] = thd;
        d[3] = thdd;
    }
}

static void integrate(const scgo_config* c, double* x, const double* u, double h, int nx) {
    double k1[12], k2[12], k3[12], k4[12], xt[12];
    f_eval(c, x, u, k1);
    if (c->euler) {
        for (int j = 0; j < nx; ++j) x[j] += h * k1[j];
        return;
    }
    for (int j = 0; j < nx; ++j) xt[j] = x[j] + h / 2 * k1[j];
    f_eval(c, xt, u, k2);
    for (int j = 0; j < nx; ++j) xt[j] = x[j] + h / 2 * k2[j];
    f_eval(c, xt, u, k3);
    for (int j = 0; j < nx; ++j) xt[j] = x[j] + h * k3[j];
    f_eval(c, xt, u, k4);
    for (int j = 0; j < nx; ++j) x[j] += h / 6 * (k1[j] + 2 * k2[j] + 2 * k3[j] + k4[j]);   /* mpc_utils.py:97 */
}

static void step_one(const scgo_config* c, const double* a_raw, double* x, int32_t* t, double* reward,
                     uint8_t* done_out, const double* ref, const double* reset_x, double* c_val, uint8_t* viol) {
    const int nx = nx_of(c->system), nu = nu_of(c->system);
    double a_clip[4], u[4];
    for (int j = 0; j < nu; ++j) {                         /* quadrotor.py:830,833 */
        double a = fmin(fmax(a_raw[j], c->act_low[j]), c->act_high[j]);
        a_clip[j] = a;
        double th = a;
        if (c->system != 4 && c->motor_map) {              /* quadrotor_utils.py:38-50,66; base_aviary.py:420 */
            int n_motor = 4 / nu;
            double pwm = (sqrt(fmax(th, 0.0) / n_motor / c->kf) - c->pwm_const) / c->pwm_scale;
            pwm = fmin(fmax(pwm, c->pwm_min), c->pwm_max);
            double rpm = c->pwm_scale * pwm + c->pwm_const;
            th = n_motor * c->kf * rpm * rpm;
        }
        u[j] = th;
    }
    const double h = c->dt / c->substeps;
    for (int s = 0; s < c->substeps; ++s) integrate(c, x, u, h, nx);
    const int tt = *t;
    const double* goal = c->x_goal;
    if (c->tracking) {
        int row = tt < c->ref_rows - 1 ? tt : c->ref_rows - 1;
        goal = ref + (int64_t)row * nx;
    }
    double e[12];
    for (int j = 0; j < nx; ++j) e[j] = x[j] - goal[j];
    int done = 0;
    if (!c->tracking && c->quadratic) {                    /* quadrotor.py:964-967 */
        double d2 = 0;
        for (int j = 0; j < nx; ++j) d2 += e[j] * e[j];
        if (sqrt(d2) < c->goal_tol) done = 1;
    }
    if (c->done_on_oob) {                                  /* :979-993 */
        for (int j = 0; j < nx; ++j)
            if (x[j] < c->oob_low[j] || x[j] > c->oob_high[j] || x[j] != x[j]) done = 1;
    }
    double l = 0;
    for (int j = 0; j < nx; ++j) l += c->w_state[j] * e[j] * e[j];
    for (int j = 0; j < nu; ++j) {
        double ea = a_clip[j] - c->u_goal[j];
        l += c->w_act[j] * ea * ea;
    }
    double rew = c->quadratic ? -0.5 * l : -l;             /* :682,909 */
    if (!c->quadratic && c->rew_exponential) rew = exp(rew);
    int violated = 0;
    for (int r = 0; r < c->n_bounds; ++r) {                /* constraints.py:301-302 */
        int dim = c->bound_dim[r];
        double v = dim < nx ? x[dim] : a_raw[dim - nx];
        double cv = c->bound_sign[r] * v - c->bound_b[r];
        if (c_val) c_val[r] = cv;
        if (cv > 0.0) violated = 1;
    }
    if (viol) *viol = (uint8_t)violated;
    if (violated && c->done_on_violation) done = 1;
    const int t1 = tt + 1;
    if (t1 >= c->ctrl_steps) done = 1;                     /* benchmark_env.py:460-462 */
    *reward = rew;
    *done_out = (uint8_t)done;
    if (done) {                                            /* dummy_vec_env.py:41-47 */
        for (int j = 0; j < nx; ++j) x[j] = reset_x[j];
        *t = 0;
    } else {
        *t = t1;
    }
}

void scgo_step(const scgo_config* c, int64_t n, const double* action, double* state, int32_t* t,
               double* reward, uint8_t* done, const double* ref, const double* reset_state, double* c_val,
               uint8_t* viol, int32_t num_threads) {
    const int nx = nx_of(c->system), nu = nu_of(c->system);
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        step_one(c, action + i * nu, state + i * nx, t + i, reward + i, done + i, ref, reset_state + i * nx,
                 c_val ? c_val + i * c->n_bounds : 0, viol ? viol + i : 0);
    }
}

int scgo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
