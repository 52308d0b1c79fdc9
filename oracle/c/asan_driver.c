/* Sanitizer driver for the C restatement (test infrastructure): runs scgo_step() on every system with exact-size
 * heap buffers -- including n = 0, n = 1 and a ragged batch, constraint rows, a reference table and auto-resets -- so that
 * AddressSanitizer / UBSan see every access the port makes.  Built by `make asan`; exit code 0 = clean. */
#include "scg_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (double)(*s >> 8) / 16777216.0; }

static int run(int system, int64_t n, int tracking, int quadratic, int n_bounds, int euler) {
    const int nx = system == 1 ? 2 : system == 2 ? 6 : system == 3 ? 12 : 4;
    const int nu = system == 1 ? 1 : system == 2 ? 2 : system == 3 ? 4 : 1;
    scgo_config c;
    memset(&c, 0, sizeof(c));
    c.system = system; c.euler = euler; c.substeps = euler ? 4 : 1; c.tracking = tracking; c.quadratic = quadratic;
    c.rew_exponential = !quadratic; c.ctrl_steps = 7; c.motor_map = system != 4; c.done_on_oob = 1; c.done_on_violation = n_bounds > 0;
    c.n_bounds = n_bounds; c.ref_rows = tracking ? 7 : 0; c.dt = 1.0 / 60;
    c.m = 0.03454; c.g = 9.8; c.l = 0.0397; c.ixx = 1.4e-5; c.iyy = 1.4e-5; c.izz = 2.17e-5; c.kf = 3.16e-10; c.km = 7.94e-12;
    c.pwm_scale = 0.2685; c.pwm_const = 4070.3; c.pwm_min = 20000; c.pwm_max = 65535; c.cp_l = 0.5; c.cp_m = 0.1; c.cp_M = 1.0;
    for (int j = 0; j < nu; ++j) { c.act_low[j] = system == 4 ? -10 : 0.028 * 4 / nu; c.act_high[j] = system == 4 ? 10 : 0.148 * 4 / nu; c.u_goal[j] = 0.0846 * 4 / nu; c.w_act[j] = 1e-4; }
    for (int j = 0; j < nx; ++j) { c.x_goal[j] = j == 0 ? 0.5 : 0.0; c.w_state[j] = 1.0; c.oob_low[j] = (j & 1) ? -3.4e38 : -2.0; c.oob_high[j] = (j & 1) ? 3.4e38 : 2.0; }
    c.goal_tol = 0.05;
    for (int k = 0; k < n_bounds; ++k) { c.bound_dim[k] = k % (nx + nu); c.bound_sign[k] = (k & 1) ? 1.0 : -1.0; c.bound_b[k] = 1.0 + 0.1 * k; }
    /* exact-size allocations: any overrun is a heap-buffer-overflow report */
    double* action = malloc(sizeof(double) * (size_t)(n * nu) + 1);
    double* state = malloc(sizeof(double) * (size_t)(n * nx) + 1);
    double* reset_state = malloc(sizeof(double) * (size_t)(n * nx) + 1);
    int32_t* t = malloc(sizeof(int32_t) * (size_t)n + 1);
    double* reward = malloc(sizeof(double) * (size_t)n + 1);
    uint8_t* done = malloc((size_t)n + 1);
    double* ref = malloc(sizeof(double) * (size_t)(c.ref_rows * nx) + 1);
    double* c_val = n_bounds ? malloc(sizeof(double) * (size_t)(n * n_bounds) + 1) : NULL;
    uint8_t* viol = n_bounds ? malloc((size_t)n + 1) : NULL;
    unsigned s = 12345u + (unsigned)system * 7u + (unsigned)n;
    for (int64_t i = 0; i < n * nx; ++i) { state[i] = 0.4 * (frand(&s) - 0.5); reset_state[i] = 0.1 * (frand(&s) - 0.5); }
    for (int64_t i = 0; i < (int64_t)c.ref_rows * nx; ++i) ref[i] = frand(&s) - 0.5;
    for (int64_t i = 0; i < n; ++i) t[i] = (int32_t)(i % 7);
    double sum = 0.0;
    for (int step = 0; step < 12; ++step) {
        for (int64_t i = 0; i < n * nu; ++i) action[i] = c.act_low[0] + (c.act_high[0] - c.act_low[0]) * (1.4 * frand(&s) - 0.2);
        scgo_step(&c, n, action, state, t, reward, done, ref, reset_state, c_val, viol, 1);
        for (int64_t i = 0; i < n; ++i) { if (!(reward[i] == reward[i])) { fprintf(stderr, "NaN reward system %d env %lld\n", system, (long long)i); return 1; } sum += reward[i] + done[i]; }
    }
    free(action); free(state); free(reset_state); free(t); free(reward); free(done); free(ref); free(c_val); free(viol);
    return sum == sum ? 0 : 1;
}

int main(void) {
    int bad = 0;
    const int64_t sizes[] = {0, 1, 3, 67};
    for (int system = 1; system <= 4; ++system)
        for (unsigned k = 0; k < sizeof(sizes) / sizeof(sizes[0]); ++k) {
            bad |= run(system, sizes[k], 0, 1, 0, 0);
            bad |= run(system, sizes[k], system != 4, 0, system == 4 ? 2 : 6, 0);
            bad |= run(system, sizes[k], 1, 1, 32, 1);
        }
    printf(bad ? "asan driver: FAILED\n" : "asan driver: ok (4 systems x 4 batch sizes x 3 configurations)\n");
    return bad;
}
