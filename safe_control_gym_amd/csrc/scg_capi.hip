// C ABI of the rollout engine (include/scg_rollout.h).  Host side only: validates arguments,
// derives the fp32 constant block from the double-precision config, owns the device-resident
// constant tables and forwards to the kernel launchers.  Never throws across the boundary.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

#include "scg_device.h"

using namespace scg;

struct scg_handle {
    scg_config cfg;
    KParams kp;
    int system, nx, nu, obs_dim;
    int device, num_cus;
    int max_lds_bytes;
    ConTable* d_con;       // device
    KTables* d_tb;         // device
    DistTable* d_dist;     // device, null without passive disturbances
    bool dist_needs_episode;   // some entry draws its step_offset per episode
    bool dist_obs;             // observation-mode entries exist
    hipStream_t aux_stream;    // second launch queue of scg_rollout (created on first use); also where rollout graphs are captured / replayed
    hipEvent_t ev_fork, ev_join;
    // hipGraph replay of small-batch rollouts (scg_config.rollout_graph): the launches of a rollout signature seen twice in a
    // row are captured once and replayed from then on.  The step serial is not baked into the nodes: they read the entry's device word.
    struct RolloutKey {
        scg_step_io io;        // reset_serial zeroed; every pointer as passed (any change is a different key)
        int64_t n, action_stride;
        int32_t steps, action_period;
    };
    struct GraphEntry {
        RolloutKey key;
        hipGraphExec_t exec;
        uint64_t last_use;
        uint32_t* d_serial;    // device word this entry's nodes read the rollout's serial base from (one per entry: two signatures
                               // replayed on two streams do not share it)
        hipEvent_t replayed;   // recorded behind every replay; waited for before the exec is destroyed
    };
    static constexpr int kGraphSlots = 4;
    GraphEntry graphs[kGraphSlots];
    RolloutKey last_key;       // signature of the previous graph-eligible call (capture happens on the second one)
    bool last_key_valid;
    bool graph_broken;         // a capture failed once: this handle stays on plain launches
    uint64_t graph_clock;
    // scg_step_async / scg_step_wait: a step issued on the handle's own queue (aux_stream) that no caller stream has joined yet
    bool async_pending;
    hipEvent_t ev_async_in, ev_async_out;
    PhysT<double> phd;
    float* d_ref;          // device, [rows][nx]
    int ref_rows;
    int n_con_rows;
};

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(SCG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                        __FILE__, __LINE__);                                                  \
    } while (0)

// One handle per device (include/scg_rollout.h, conventions): the handle's tables and queues live on the device that was current
// in scg_create(); a call made with another device current would launch against them from the wrong device.
#define GUARD_DEVICE(h, who)                                                                                   \
    do {                                                                                                       \
        int _dev = -1;                                                                                         \
        if (hipGetDevice(&_dev) != hipSuccess || _dev != (h)->device)                                          \
            return fail(SCG_ERR_INVALID, "%s: the handle belongs to device %d but device %d is current "      \
                        "(one handle per device; make the call with the handle's device current)", who, (h)->device, _dev); \
    } while (0)

static bool dims_of(int system, int& nx, int& nu) {
    switch (system) {
        case SCG_QUAD_1D: nx = 2; nu = 1; return true;
        case SCG_QUAD_2D: nx = 6; nu = 2; return true;
        case SCG_QUAD_3D: nx = 12; nu = 4; return true;
        case SCG_CARTPOLE: nx = 4; nu = 1; return true;
        default: return false;
    }
}

static float f32_clamped(double v) {
    // +-inf bounds stay infinite; finite doubles beyond float range saturate (f32max bounds of
    // quadrotor.py:752-785 are exactly representable).
    if (std::isinf(v)) return v > 0 ? std::numeric_limits<float>::infinity() : -std::numeric_limits<float>::infinity();
    const double m = (double)std::numeric_limits<float>::max();
    if (v > m) return std::numeric_limits<float>::max();
    if (v < -m) return -std::numeric_limits<float>::max();
    return (float)v;
}

static bool matrix_is_diag(const double* M, int n, int stride) {
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b)
            if (a != b && M[a * stride + b] != 0.0) return false;
    return true;
}

static void derive_params(scg_handle* h) {
    const scg_config& c = h->cfg;
    KParams& k = h->kp;
    std::memset(&k, 0, sizeof(k));
    const int nx = h->nx, nu = h->nu;
    uint32_t f = 0;
    if (c.normalized_action) f |= F_NORMALIZED;
    if (c.motor_map && c.system != SCG_CARTPOLE) f |= F_MOTOR_MAP;
    if (c.rew_exponential) f |= F_REW_EXP;
    if (c.done_on_out_of_bound) f |= F_DONE_OOB;
    if (c.done_on_violation) f |= F_DONE_VIOL;
    if (c.use_constraint_penalty) f |= F_PENALTY;
    if (c.randomized_init) f |= F_RAND_INIT;
    if (c.auto_reset) f |= F_AUTO_RESET;
    if (c.randomized_inertial) f |= F_RAND_INERTIAL;
    if (c.task == SCG_TRAJ_TRACKING) f |= F_TRACKING;
    if (c.cost == SCG_QUADRATIC) f |= F_QUADRATIC;
    if (c.integrator == SCG_EULER) f |= F_EULER;
    if (c.ground_effect && c.system != SCG_CARTPOLE) f |= F_GND;
    if (c.drag && c.system != SCG_CARTPOLE) f |= F_DRAG;
    if (c.cost != SCG_QUADRATIC || (matrix_is_diag(c.Q, nx, SCG_MAX_NX) && matrix_is_diag(c.R, nu, SCG_MAX_NU)))
        f |= F_Q_DIAG;
    if (c.randomized_init)
        for (int j = 0; j < nx; ++j) if (c.init_rand_kind[j] != SCG_RAND_UNIFORM) f |= F_RAND_GENERAL;
    if (c.randomized_inertial)
        for (int q = 0; q < 4; ++q) if (c.inertial_rand_kind[q] != SCG_RAND_UNIFORM) f |= F_RAND_GENERAL;
    if (f & F_Q_DIAG) {
        bool unit = true;
        for (int j = 0; j < nx; ++j)
            if ((float)(c.cost == SCG_QUADRATIC ? c.Q[j * SCG_MAX_NX + j] : c.rew_state_weight[j]) != 1.0f) unit = false;
        if (unit) f |= F_W_STATE_UNIT;
    }
    k.flags = f;
    k.substeps = c.substeps;
    k.ctrl_steps = c.ctrl_steps;
    k.ref_rows = 0;
    k.obs_goal_horizon = (c.cost == SCG_RL_REWARD) ? c.obs_goal_horizon : 0;
    if (c.task == SCG_STABILIZATION && k.obs_goal_horizon > 1) k.obs_goal_horizon = 1;
    k.n_con_rows = 0;
    k.grid_cap_per_cu = c.grid_cap_per_cu > 0 ? c.grid_cap_per_cu : 0;
    // dims whose out-of-bound box is binding: the reference bounds the velocities / rates with +-f32max
    // (quadrotor.py:742-791), which no finite float violates -- the kernel skips those compares.  A non-finite
    // velocity still ends the episode: it reaches the position (a bounded dim) within the same RK4 step.
    k.oob_dims = 0;
    k.oob_sym = 0;
    for (int j = 0; j < nx; ++j) {
        const double m = (double)std::numeric_limits<float>::max();
        if (c.oob_low[j] > -m || c.oob_high[j] < m) k.oob_dims |= 1u << j;
        if (f32_clamped(c.oob_low[j]) == -f32_clamped(c.oob_high[j])) k.oob_sym |= 1u << j;
    }
    // explicit Euler advances the positions with the OLD velocities: a non-finite velocity reaches a bounded dim one step
    // late, so there every dim keeps its compare (inf > f32max, as in the reference)
    if (c.integrator == SCG_EULER) k.oob_dims = (1u << nx) - 1u;
    k.seed_lo = (uint32_t)(c.seed & 0xffffffffull);
    k.seed_hi = (uint32_t)(c.seed >> 32);
    k.env_base = c.env_index_base;
    PhysT<double> d;
    std::memset(&d, 0, sizeof(d));
    d.h = c.dt / (double)c.substeps;
    if (c.system != SCG_CARTPOLE) {
        const double n_motor = 4.0 / nu;
        d.inv_m = 1.0 / c.mass;
        d.g = c.gravity;
        d.arm_s2 = c.arm / std::sqrt(2.0);
        d.gamma = c.km / c.kf;
        d.inv_ixx = 1.0 / c.ixx;
        d.inv_iyy = 1.0 / c.iyy;
        d.inv_izz = 1.0 / c.izz;
        d.cxx = (c.izz - c.iyy) / c.ixx;
        d.cyy = (c.ixx - c.izz) / c.iyy;
        d.czz = (c.iyy - c.ixx) / c.izz;
        d.l_s2_iyy = c.arm / c.iyy / std::sqrt(2.0);
        d.inv_nmotor_kf = 1.0 / (n_motor * c.kf);
        d.nmotor_kf = n_motor * c.kf;
        d.pwm_scale = c.pwm2rpm_scale;
        d.inv_pwm_scale = 1.0 / c.pwm2rpm_scale;
        d.pwm_const = c.pwm2rpm_const;
        d.pwm_min = c.pwm_min;
        d.pwm_max = c.pwm_max;
        const double rpm_min = c.pwm2rpm_scale * c.pwm_min + c.pwm2rpm_const, rpm_max = c.pwm2rpm_scale * c.pwm_max + c.pwm2rpm_const;
        d.th_min = n_motor * c.kf * rpm_min * rpm_min;
        d.th_max = n_motor * c.kf * rpm_max * rpm_max;
    } else {
        d.cp_l = c.pole_length;
        d.cp_m = c.pole_mass;
        d.cp_inv_Mm = 1.0 / (c.pole_mass + c.cart_mass);
        d.cp_ml = c.pole_mass * c.pole_length;
        d.cp_g = c.gravity;
    }
    h->phd = d;
    {
        // fp32 copy, member by member (PhysT<float> has the same field order)
        const double* src = reinterpret_cast<const double*>(&d);
        float* dst = reinterpret_cast<float*>(&k.ph);
        for (size_t q = 0; q < sizeof(d) / sizeof(double); ++q) dst[q] = (float)src[q];
    }
    k.act_mul = 1.0f;
    k.act_add = 0.0f;
    if (c.normalized_action) {
        if (c.system == SCG_CARTPOLE) {
            k.act_mul = (float)c.action_scale;
        } else {
            k.act_mul = (float)(c.norm_act_scale * c.hover_thrust);
            k.act_add = (float)c.hover_thrust;
        }
    }
    k.goal_tol2 = (float)(c.goal_tolerance * c.goal_tolerance);
    k.constraint_penalty = (float)c.constraint_penalty;
    k.cost_scale = (c.cost == SCG_QUADRATIC) ? -0.5f : -1.0f;
    k.gnd_k = (float)(c.gnd_eff_coeff * (c.prop_radius / 4.0) * (c.prop_radius / 4.0));
    k.gnd_h_clip = (float)c.gnd_eff_h_clip;
    k.drag_xy = (float)(c.drag_coeff_xy * 2.0 * M_PI / 60.0);
    k.drag_z = (float)(c.drag_coeff_z * 2.0 * M_PI / 60.0);
}

static void derive_tables(const scg_handle* h, KTables& t) {
    const scg_config& c = h->cfg;
    std::memset(&t, 0, sizeof(t));
    const int nx = h->nx, nu = h->nu;
    const bool quad_cost = (c.cost == SCG_QUADRATIC);
    for (int j = 0; j < nu; ++j) {
        t.act_low[j] = f32_clamped(c.act_low[j]);
        t.act_high[j] = f32_clamped(c.act_high[j]);
        t.u_goal[j] = (float)c.u_goal[j];
        t.w_act[j] = (float)(quad_cost ? c.R[j * SCG_MAX_NU + j] : c.rew_act_weight[j]);
    }
    for (int j = 0; j < nx; ++j) {
        t.x_goal[j] = (float)c.x_goal[j];
        t.w_state[j] = (float)(quad_cost ? c.Q[j * SCG_MAX_NX + j] : c.rew_state_weight[j]);
        t.mse_w[j] = (float)c.mse_weight[j];
        t.oob_low[j] = f32_clamped(c.oob_low[j]);
        t.oob_high[j] = f32_clamped(c.oob_high[j]);
        t.init_state[j] = (float)c.init_state[j];
        t.init_lo[j] = (float)c.init_rand_low[j];
        t.init_span[j] = (float)(c.init_rand_high[j] - c.init_rand_low[j]);
    }
    t.phd = h->phd;
    t.arm = c.arm;
    t.kmkf = c.system != SCG_CARTPOLE ? c.km / c.kf : 0.0;
    const double nominal[4] = {c.system == SCG_CARTPOLE ? c.pole_length : c.mass,
                               c.system == SCG_CARTPOLE ? c.pole_mass : c.ixx,
                               c.system == SCG_CARTPOLE ? c.cart_mass : c.iyy,
                               c.system == SCG_CARTPOLE ? 0.0 : c.izz};
    for (int q = 0; q < 4; ++q) {
        t.inertial_nominal[q] = (float)nominal[q];
        t.inertial_lo[q] = (float)c.inertial_rand_low[q];
        t.inertial_span[q] = (float)(c.inertial_rand_high[q] - c.inertial_rand_low[q]);
    }
    for (int j = 0; j < nu; ++j) { t.act_low_d[j] = c.act_low[j]; t.act_high_d[j] = c.act_high[j]; }
    for (int j = 0; j < nx; ++j) {
        t.init_kind[j] = c.init_rand_kind[j];
        for (int r = 0; r < 3; ++r) t.init_p[j][r] = (float)c.init_rand_p[j][r];
    }
    for (int q = 0; q < 4; ++q) {
        t.inertial_kind[q] = c.inertial_rand_kind[q];
        for (int r = 0; r < 3; ++r) t.inertial_p[q][r] = (float)c.inertial_rand_p[q][r];
    }
    for (int a = 0; a < SCG_MAX_NX * SCG_MAX_NX; ++a) t.Q[a] = (float)c.Q[a];
    for (int a = 0; a < SCG_MAX_NU * SCG_MAX_NU; ++a) t.R[a] = (float)c.R[a];
}

// Every instantiated rollout graph bakes the handle's table pointers and constants into its nodes: whatever changes those
// (scg_set_reference / _constraints / _disturbances) drops the graphs; they are captured again on demand.
// An exec may still be replaying when it has to go: every replay is followed by an event (GraphEntry::replayed) and the exec is
// destroyed only after that event has completed.  If the wait itself fails (e.g. another thread holds a global-mode capture) the
// exec is LEAKED rather than destroyed under a running replay.
static void retire_graph_entry(scg_handle::GraphEntry& g) {
    if (g.exec) {
        const bool idle = !g.replayed || hipEventSynchronize(g.replayed) == hipSuccess;
        if (idle) (void)hipGraphExecDestroy(g.exec);
        else (void)hipGetLastError();
    }
    g.exec = nullptr;
}

static void drop_rollout_graphs(scg_handle* h) {
    for (int q = 0; q < scg_handle::kGraphSlots; ++q) retire_graph_entry(h->graphs[q]);
    h->last_key_valid = false;
}

// The handle's own launch queue (second queue of scg_rollout, capture / home queue of scg_step_async) and its events, created on
// first use.
static int ensure_aux_queue(scg_handle* h, const char* who) {
    if (h->aux_stream) return SCG_OK;
    hipStream_t s = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int q = 0; q < 4 && e == hipSuccess; ++q) e = hipEventCreateWithFlags(&ev[q], hipEventDisableTiming);
    if (e != hipSuccess) {
        for (int q = 3; q >= 0; --q) if (ev[q]) (void)hipEventDestroy(ev[q]);
        if (s) (void)hipStreamDestroy(s);
        return fail(SCG_ERR_HIP, "%s: the handle's launch queue: %s", who, hipGetErrorString(e));
    }
    h->aux_stream = s; h->ev_fork = ev[0]; h->ev_join = ev[1]; h->ev_async_in = ev[2]; h->ev_async_out = ev[3];
    return SCG_OK;
}

// A step issued by scg_step_async runs on the handle's own queue; before anything else of this handle is launched on a caller's
// stream, that stream is made to wait for it (device-side; a no-op without a pending step or on the handle's queue itself).
static int join_async(scg_handle* h, hipStream_t stream, const char* who) {
    if (!h->async_pending || stream == h->aux_stream) return SCG_OK;
    hipError_t e = hipEventRecord(h->ev_async_out, h->aux_stream);      // everything queued on the handle's queue so far
    if (e == hipSuccess) e = hipStreamWaitEvent(stream, h->ev_async_out, 0);
    if (e != hipSuccess) return fail(SCG_ERR_HIP, "%s: joining the pending asynchronous step failed: %s", who, hipGetErrorString(e));
    return SCG_OK;
}

extern "C" {

int scg_version(void) { return SCG_ABI_VERSION; }

const char* scg_last_error(void) { return g_err; }

int scg_sizeof_config(void) { return (int)sizeof(scg_config); }
int scg_sizeof_step_io(void) { return (int)sizeof(scg_step_io); }
int scg_sizeof_constraint_row(void) { return (int)sizeof(scg_constraint_row); }
int scg_sizeof_disturbance(void) { return (int)sizeof(scg_disturbance); }
int scg_sizeof_trajectory_sink(void) { return (int)sizeof(scg_trajectory_sink); }

int scg_create(const scg_config* cfg, scg_handle** out) {
    if (!cfg || !out) return fail(SCG_ERR_INVALID, "scg_create: null argument");
    if (cfg->struct_size != sizeof(scg_config))
        return fail(SCG_ERR_INVALID, "scg_create: struct_size %u != %zu", cfg->struct_size, sizeof(scg_config));
    if (cfg->abi_version != SCG_ABI_VERSION)
        return fail(SCG_ERR_INVALID, "scg_create: abi_version %u != %d", cfg->abi_version, SCG_ABI_VERSION);
    int nx, nu;
    if (!dims_of(cfg->system, nx, nu)) return fail(SCG_ERR_INVALID, "scg_create: unknown system %d", cfg->system);
    if (cfg->integrator != SCG_RK4 && cfg->integrator != SCG_EULER)
        return fail(SCG_ERR_INVALID, "scg_create: unknown integrator %d", cfg->integrator);
    if (cfg->task != SCG_STABILIZATION && cfg->task != SCG_TRAJ_TRACKING)
        return fail(SCG_ERR_INVALID, "scg_create: unknown task %d", cfg->task);
    if (cfg->cost != SCG_RL_REWARD && cfg->cost != SCG_QUADRATIC)
        return fail(SCG_ERR_UNSUPPORTED, "scg_create: cost %d is not on the hot path", cfg->cost);
    if (cfg->substeps < 1 || cfg->substeps > 64) return fail(SCG_ERR_INVALID, "scg_create: substeps %d out of [1,64]", cfg->substeps);
    if (!(cfg->dt > 0.0)) return fail(SCG_ERR_INVALID, "scg_create: dt must be positive");
    if (cfg->ctrl_steps < 1) return fail(SCG_ERR_INVALID, "scg_create: ctrl_steps must be >= 1");
    if (cfg->obs_goal_horizon < 0 || cfg->obs_goal_horizon > SCG_MAX_GOAL_HORIZON)
        return fail(SCG_ERR_INVALID, "scg_create: obs_goal_horizon %d out of [0,%d]", cfg->obs_goal_horizon, SCG_MAX_GOAL_HORIZON);
    if (cfg->system != SCG_CARTPOLE) {
        if (!(cfg->mass > 0 && cfg->ixx > 0 && cfg->iyy > 0 && cfg->izz > 0 && cfg->kf > 0 && cfg->pwm2rpm_scale > 0))
            return fail(SCG_ERR_INVALID, "scg_create: non-positive physical constant");
    } else if (!(cfg->pole_length > 0 && cfg->pole_mass > 0 && cfg->cart_mass > 0)) {
        return fail(SCG_ERR_INVALID, "scg_create: non-positive cartpole constant");
    }

    if (cfg->rollout_queues < 0 || cfg->rollout_queues > 2)
        return fail(SCG_ERR_INVALID, "scg_create: rollout_queues %d out of {0 (default), 1, 2}", cfg->rollout_queues);
    if (cfg->rollout_graph < -1 || cfg->rollout_graph > 1)
        return fail(SCG_ERR_INVALID, "scg_create: rollout_graph %d out of {-1 (off), 0 (default: on), 1 (on)}", cfg->rollout_graph);
    if (cfg->lone_launch_forms < -1 || cfg->lone_launch_forms > 1)
        return fail(SCG_ERR_INVALID, "scg_create: lone_launch_forms %d out of {-1 (off), 0 (default: on), 1 (on)}", cfg->lone_launch_forms);
    if (cfg->rollout_split_min < 0 || cfg->grid_cap_per_cu < 0)
        return fail(SCG_ERR_INVALID, "scg_create: rollout_split_min and grid_cap_per_cu must be >= 0");
    for (int j = 0; j < SCG_MAX_NX; ++j)
        if (cfg->init_rand_kind[j] < SCG_RAND_UNIFORM || cfg->init_rand_kind[j] > SCG_RAND_KIND_MAX)
            return fail(SCG_ERR_INVALID, "scg_create: init_rand_kind[%d] = %d is not a scg_distribution", j, cfg->init_rand_kind[j]);
    for (int q = 0; q < 4; ++q)
        if (cfg->inertial_rand_kind[q] < SCG_RAND_UNIFORM || cfg->inertial_rand_kind[q] > SCG_RAND_KIND_MAX)
            return fail(SCG_ERR_INVALID, "scg_create: inertial_rand_kind[%d] = %d is not a scg_distribution", q, cfg->inertial_rand_kind[q]);

    // parameter ranges of the rejection samplers and of the weighted choice (the kernel divides by / takes roots of them)
    auto bad_params = [](int kind, const double* prm, double low, double high) -> const char* {
        if (kind == SCG_RAND_GAMMA && !(prm[0] > 0.0 && prm[1] >= 0.0)) return "gamma needs shape > 0 and scale >= 0";
        if (kind == SCG_RAND_BETA && !(prm[0] > 0.0 && prm[1] > 0.0)) return "beta needs a > 0 and b > 0";
        if (kind == SCG_RAND_CHOICE_P && !(low >= 0.0 && high >= low && high <= 1.0 + 1e-9)) return "weighted choice needs 0 <= P(p0) <= P(p0) + P(p1) <= 1 in rand_low / rand_high";
        return nullptr;
    };
    for (int j = 0; j < nx && cfg->randomized_init; ++j)
        if (const char* why = bad_params(cfg->init_rand_kind[j], cfg->init_rand_p[j], cfg->init_rand_low[j], cfg->init_rand_high[j]))
            return fail(SCG_ERR_INVALID, "scg_create: init_rand_kind[%d]: %s", j, why);
    for (int q = 0; q < 4 && cfg->randomized_inertial; ++q)
        if (const char* why = bad_params(cfg->inertial_rand_kind[q], cfg->inertial_rand_p[q], cfg->inertial_rand_low[q], cfg->inertial_rand_high[q]))
            return fail(SCG_ERR_INVALID, "scg_create: inertial_rand_kind[%d]: %s", q, why);
    // SCG_RAND_CHOICE carries its number of values in *_rand_low: 1, 2 or 3 (the kernel selects among p[0..2])
    for (int j = 0; j < nx; ++j)
        if (cfg->randomized_init && cfg->init_rand_kind[j] == SCG_RAND_CHOICE &&
            !(cfg->init_rand_low[j] == 1.0 || cfg->init_rand_low[j] == 2.0 || cfg->init_rand_low[j] == 3.0))
            return fail(SCG_ERR_INVALID, "scg_create: init_rand_kind[%d] = choice needs a value count of 1, 2 or 3 in init_rand_low (got %g)",
                        j, cfg->init_rand_low[j]);
    for (int q = 0; q < 4; ++q)
        if (cfg->randomized_inertial && cfg->inertial_rand_kind[q] == SCG_RAND_CHOICE &&
            !(cfg->inertial_rand_low[q] == 1.0 || cfg->inertial_rand_low[q] == 2.0 || cfg->inertial_rand_low[q] == 3.0))
            return fail(SCG_ERR_INVALID, "scg_create: inertial_rand_kind[%d] = choice needs a value count of 1, 2 or 3 in inertial_rand_low (got %g)",
                        q, cfg->inertial_rand_low[q]);

    if (cfg->randomized_inertial) {
        const double nominal[4] = {cfg->system == SCG_CARTPOLE ? cfg->pole_length : cfg->mass,
                                   cfg->system == SCG_CARTPOLE ? cfg->pole_mass : cfg->ixx,
                                   cfg->system == SCG_CARTPOLE ? cfg->cart_mass : cfg->iyy,
                                   cfg->system == SCG_CARTPOLE ? 1.0 : cfg->izz};
        for (int q = 0; q < 4; ++q) {
            // smallest draw of a bounded distribution (unbounded ones are floored by the kernel at 0.1 % of nominal)
            const int kind = cfg->inertial_rand_kind[q];
            double lowest = 0.0;
            bool bounded = true;
            if (kind == SCG_RAND_UNIFORM) {
                lowest = cfg->inertial_rand_low[q];
                if (cfg->inertial_rand_high[q] < cfg->inertial_rand_low[q])
                    return fail(SCG_ERR_INVALID, "scg_create: randomised inertial property %d has high < low", q);
            } else if (kind == SCG_RAND_TRIANGULAR || kind == SCG_RAND_RANDINT) lowest = cfg->inertial_rand_p[q][0];
            else if (kind == SCG_RAND_EXPONENTIAL || kind == SCG_RAND_RAYLEIGH || kind == SCG_RAND_WEIBULL || kind == SCG_RAND_PARETO ||
                     kind == SCG_RAND_POWER || kind == SCG_RAND_LOGNORMAL || kind == SCG_RAND_GAMMA || kind == SCG_RAND_BETA) lowest = 0.0;
            else if (kind == SCG_RAND_CHOICE_P) {
                // values with a non-zero probability: p0 if low > 0, p1 if high > low, p2 if high < 1
                const double lo = cfg->inertial_rand_low[q], hi = cfg->inertial_rand_high[q];
                lowest = 1e300;
                if (lo > 0.0) lowest = cfg->inertial_rand_p[q][0];
                if (hi > lo && cfg->inertial_rand_p[q][1] < lowest) lowest = cfg->inertial_rand_p[q][1];
                if (hi < 1.0 && cfg->inertial_rand_p[q][2] < lowest) lowest = cfg->inertial_rand_p[q][2];
            }
            else if (kind == SCG_RAND_CHOICE) {
                const int cnt = (int)cfg->inertial_rand_low[q];
                lowest = cfg->inertial_rand_p[q][0];
                for (int r = 1; r < cnt && r < 3; ++r) lowest = cfg->inertial_rand_p[q][r] < lowest ? cfg->inertial_rand_p[q][r] : lowest;
            } else bounded = false;
            if (bounded && !(nominal[q] + lowest > 0.0))
                return fail(SCG_ERR_INVALID, "scg_create: randomised inertial property %d can become non-positive", q);
        }
    }

    scg_handle* h = new (std::nothrow) scg_handle();
    if (!h) return fail(SCG_ERR_NOMEM, "scg_create: out of host memory");
    std::memset(h, 0, sizeof(*h));
    h->cfg = *cfg;
    h->system = cfg->system;
    h->nx = nx;
    h->nu = nu;
    hipError_t e = hipGetDevice(&h->device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, h->device);
    if (e != hipSuccess) {
        delete h;
        return fail(SCG_ERR_HIP, "scg_create: no usable HIP device: %s", hipGetErrorString(e));
    }
    h->num_cus = prop.multiProcessorCount;
    h->max_lds_bytes = (int)prop.sharedMemPerBlock;
    derive_params(h);
    h->obs_dim = nx * (1 + h->kp.obs_goal_horizon);

    // constant tables: the per-dimension vectors and an empty constraint table
    std::vector<char> zero(sizeof(ConTable), 0);
    ConTable* ct = reinterpret_cast<ConTable*>(zero.data());
    ct->n_state_dims = nx;
    KTables tb;
    derive_tables(h, tb);
    e = hipMalloc(&h->d_con, sizeof(ConTable));
    if (e == hipSuccess) e = hipMalloc(&h->d_tb, sizeof(KTables));
    if (e == hipSuccess) e = hipMemcpy(h->d_con, ct, sizeof(ConTable), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_tb, &tb, sizeof(KTables), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (h->d_con) (void)hipFree(h->d_con);
        if (h->d_tb) (void)hipFree(h->d_tb);
        delete h;
        return fail(SCG_ERR_HIP, "scg_create: device table setup failed: %s", hipGetErrorString(e));
    }
    *out = h;
    return SCG_OK;
}

int scg_destroy(scg_handle* h) {
    if (!h) return SCG_OK;
    if (h->aux_stream) (void)hipStreamSynchronize(h->aux_stream);   // a step issued by scg_step_async may still be reading the tables
    if (h->d_con) (void)hipFree(h->d_con);
    if (h->d_tb) (void)hipFree(h->d_tb);
    if (h->d_ref) (void)hipFree(h->d_ref);
    if (h->d_dist) (void)hipFree(h->d_dist);
    drop_rollout_graphs(h);
    for (int q = 0; q < scg_handle::kGraphSlots; ++q) {
        if (h->graphs[q].d_serial) (void)hipFree(h->graphs[q].d_serial);
        if (h->graphs[q].replayed) (void)hipEventDestroy(h->graphs[q].replayed);
    }
    if (h->aux_stream) {
        (void)hipEventDestroy(h->ev_fork); (void)hipEventDestroy(h->ev_join);
        (void)hipEventDestroy(h->ev_async_in); (void)hipEventDestroy(h->ev_async_out);
        (void)hipStreamDestroy(h->aux_stream);
    }
    delete h;
    return SCG_OK;
}

int scg_set_disturbances(scg_handle* h, const scg_disturbance* entries, int32_t n) {
    if (!h) return fail(SCG_ERR_INVALID, "scg_set_disturbances: null handle");
    GUARD_DEVICE(h, "scg_set_disturbances");
    if (n < 0 || (n > 0 && !entries)) return fail(SCG_ERR_INVALID, "scg_set_disturbances: bad entry list");
    DistTable t;
    std::memset(&t, 0, sizeof(t));
    bool needs_episode = false;
    const int dyn_dim = h->system == SCG_QUAD_3D ? 3 : (h->system == SCG_QUAD_2D ? 2 : 1);
    const int dims[3] = {h->nu, dyn_dim, h->nx};
    for (int32_t k = 0; k < n; ++k) {
        const scg_disturbance& s = entries[k];
        if (s.mode < 0 || s.mode > 2) return fail(SCG_ERR_INVALID, "scg_set_disturbances: entry %d: mode %d", k, s.mode);
        if (s.kind < SCG_DIST_IMPULSE || s.kind > SCG_DIST_PERIODIC)
            return fail(SCG_ERR_INVALID, "scg_set_disturbances: entry %d: kind %d", k, s.kind);
        if (t.n[s.mode] >= SCG_MAX_DISTURBANCES)
            return fail(SCG_ERR_INVALID, "scg_set_disturbances: more than %d entries for mode %d", SCG_MAX_DISTURBANCES, s.mode);
        DistEntry& d = t.e[s.mode][t.n[s.mode]++];
        d.kind = s.kind;
        const int dim = dims[s.mode];
        for (int j = 0; j < SCG_MAX_NX; ++j) d.mask[j] = (j < dim) ? (s.has_mask ? (float)s.mask[j] : 1.0f) : 0.0f;
        if (s.kind == SCG_DIST_IMPULSE || s.kind == SCG_DIST_STEP) {
            if (s.step_offset < 0 && s.max_step < 1)
                return fail(SCG_ERR_INVALID, "scg_set_disturbances: entry %d: random step_offset needs max_step >= 1", k);
            d.step_offset = s.step_offset < 0 ? -1 : s.step_offset;
            d.max_step = s.max_step;
            d.magnitude = (float)s.magnitude;
            if (s.step_offset < 0) needs_episode = true;
            if (s.kind == SCG_DIST_IMPULSE) {
                // disturbances.py:119-120: duration >= 1, 0 < decay_rate <= 1
                if (s.duration < 1 || !(s.decay_rate > 0.0 && s.decay_rate <= 1.0))
                    return fail(SCG_ERR_INVALID, "scg_set_disturbances: entry %d: impulse needs duration >= 1 and 0 < decay_rate <= 1", k);
                d.duration = s.duration;
                d.half_duration = (float)(s.duration / 2.0);
                d.log2_decay = (float)std::log2(s.decay_rate);
            }
        } else if (s.kind == SCG_DIST_UNIFORM) {
            for (int j = 0; j < dim; ++j) { d.a[j] = (float)s.low[j]; d.b[j] = (float)(s.high[j] - s.low[j]); }
        } else if (s.kind == SCG_DIST_WHITE_NOISE) {
            for (int j = 0; j < dim; ++j) d.a[j] = (float)s.low[j];
        } else {
            d.scale = (float)s.scale;
            d.omega_dt = (float)(2.0 * M_PI * s.frequency * h->cfg.dt);
        }
    }
    drop_rollout_graphs(h);
    if (h->d_dist) { (void)hipFree(h->d_dist); h->d_dist = nullptr; }
    h->kp.dist = nullptr;
    h->dist_needs_episode = false;
    h->dist_obs = false;
    if (n == 0) return SCG_OK;
    HIP_TRY(hipMalloc(&h->d_dist, sizeof(DistTable)));
    HIP_TRY(hipMemcpy(h->d_dist, &t, sizeof(DistTable), hipMemcpyHostToDevice));
    h->kp.dist = h->d_dist;
    h->dist_needs_episode = needs_episode;
    h->dist_obs = t.n[SCG_DIST_OBSERVATION] > 0;
    return SCG_OK;
}

// buffers the attached disturbances need (shared by scg_step / scg_rollout / scg_rollout_fused)
static int check_dist_io(const scg_handle* h, const scg_step_io* io, const char* who) {
    const bool obs_is_state = io->obs && (io->obs == io->state || io->obs == io->state_out);
    if (h->dist_obs && (!io->obs || obs_is_state))
        return fail(SCG_ERR_INVALID, "%s: an observation disturbance needs an obs buffer of its own (obs must not alias state)", who);
    if (io->obs_noise && obs_is_state)
        return fail(SCG_ERR_INVALID, "%s: obs_noise needs an obs buffer of its own (obs must not alias state)", who);
    if (io->obs && io->state_out && io->state_out != io->state && io->obs == io->state)
        return fail(SCG_ERR_INVALID, "%s: with state_out, a shared obs buffer must be the one the state is written to (obs == state_out)", who);
    if (h->dist_needs_episode && !io->episode_id)
        return fail(SCG_ERR_INVALID, "%s: a per-episode step_offset needs the episode_id plane", who);
    return SCG_OK;
}

int scg_set_reference(scg_handle* h, const double* x_goal, int32_t rows) {
    if (!h || !x_goal) return fail(SCG_ERR_INVALID, "scg_set_reference: null argument");
    GUARD_DEVICE(h, "scg_set_reference");
    if (rows < 1) return fail(SCG_ERR_INVALID, "scg_set_reference: rows must be >= 1");
    if (h->cfg.task != SCG_TRAJ_TRACKING) return fail(SCG_ERR_INVALID, "scg_set_reference: task is not traj_tracking");
    if (h->cfg.cost == SCG_QUADRATIC && rows < h->cfg.ctrl_steps)
        return fail(SCG_ERR_INVALID, "scg_set_reference: quadratic cost indexes X_GOAL[t] for t < %d but only %d rows given",
                    h->cfg.ctrl_steps, rows);
    const size_t count = (size_t)rows * h->nx;
    std::vector<float> tmp(count);
    for (size_t a = 0; a < count; ++a) tmp[a] = (float)x_goal[a];
    drop_rollout_graphs(h);
    if (h->d_ref) { (void)hipFree(h->d_ref); h->d_ref = nullptr; }
    // padded to a multiple of 16 bytes: the LDS staging loop copies float4
    const size_t bytes = ((count * sizeof(float) + 15) / 16) * 16;
    HIP_TRY(hipMalloc(&h->d_ref, bytes));
    HIP_TRY(hipMemset(h->d_ref, 0, bytes));
    HIP_TRY(hipMemcpy(h->d_ref, tmp.data(), count * sizeof(float), hipMemcpyHostToDevice));
    h->ref_rows = rows;
    h->kp.ref_rows = rows;
    // Reference-table placement.  Default: gathered from global memory through L1/L2 (measured faster
    // at the default table size).  scg_config.ref_in_lds stages it in LDS instead
    // (when it leaves room for >= 2 workgroups per CU).
    const bool want_lds = h->cfg.ref_in_lds != 0;
    const int lds_budget = h->max_lds_bytes > 0 ? h->max_lds_bytes / 2 : 32768;
    if (want_lds && (int64_t)bytes <= lds_budget) h->kp.flags |= F_REF_IN_LDS; else h->kp.flags &= ~F_REF_IN_LDS;
    return SCG_OK;
}

int scg_set_constraints(scg_handle* h, const scg_constraint_row* rows, int32_t n_rows, const double* p_mats,
                        int32_t n_p) {
    if (!h) return fail(SCG_ERR_INVALID, "scg_set_constraints: null handle");
    GUARD_DEVICE(h, "scg_set_constraints");
    if (n_rows < 0 || n_rows > SCG_MAX_CONSTRAINT_ROWS)
        return fail(SCG_ERR_INVALID, "scg_set_constraints: %d rows (max %d)", n_rows, SCG_MAX_CONSTRAINT_ROWS);
    if (n_rows > 0 && !rows) return fail(SCG_ERR_INVALID, "scg_set_constraints: null rows");
    if (n_p < 0 || n_p > SCG_MAX_QUADRATIC) return fail(SCG_ERR_INVALID, "scg_set_constraints: %d quadratic forms (max %d)", n_p, SCG_MAX_QUADRATIC);
    drop_rollout_graphs(h);
    const int nv = h->nx + h->nu;
    std::vector<char> buf(sizeof(ConTable), 0);
    ConTable* ct = reinterpret_cast<ConTable*>(buf.data());
    ct->n_rows = n_rows;
    ct->n_state_dims = h->nx;
    // classify: a linear row with a single +-1 coefficient is a bound row
    std::vector<int> bound_dim(n_rows, -1);
    for (int r = 0; r < n_rows; ++r) {
        const scg_constraint_row& row = rows[r];
        if (row.kind == 1) {
            if (row.p_index < 0 || row.p_index >= n_p || !p_mats)
                return fail(SCG_ERR_INVALID, "scg_set_constraints: row %d has a bad p_index", r);
            continue;
        }
        if (row.kind != 0) return fail(SCG_ERR_INVALID, "scg_set_constraints: row %d has unknown kind %d", r, row.kind);
        int nnz = 0, dim = -1;
        for (int k = 0; k < SCG_MAX_NV; ++k) {
            if (row.coef[k] != 0.0) {
                if (k >= nv) return fail(SCG_ERR_INVALID, "scg_set_constraints: row %d reads v[%d] but nv = %d", r, k, nv);
                ++nnz; dim = k;
            }
        }
        if (nnz == 1 && (row.coef[dim] == 1.0 || row.coef[dim] == -1.0)) bound_dim[r] = dim;
    }
    int nb = 0;
    for (int k = 0; k < SCG_MAX_NV; ++k) {
        ct->bound_start[k] = nb;
        for (int r = 0; r < n_rows; ++r) {
            if (rows[r].kind == 0 && bound_dim[r] == k) {
                BoundEntry& be = ct->bound[nb++];
                be.sign = (float)rows[r].coef[k];
                be.b = f32_clamped(rows[r].b);
                be.tol = (float)rows[r].tolerance;
                be.out_row = r;
                be.flags = (rows[r].strict ? RF_STRICT : 0) | (rows[r].has_tolerance ? RF_HAS_TOL : 0);
            }
        }
    }
    ct->bound_start[SCG_MAX_NV] = nb;
    ct->n_bound = nb;
    int nd = 0, nq = 0;
    for (int r = 0; r < n_rows; ++r) {
        const scg_constraint_row& row = rows[r];
        const int flags = (row.strict ? RF_STRICT : 0) | (row.has_tolerance ? RF_HAS_TOL : 0);
        if (row.kind == 0 && bound_dim[r] < 0) {
            DenseRow& dr = ct->dense[nd++];
            for (int k = 0; k < SCG_MAX_NV; ++k) dr.coef[k] = (float)row.coef[k];
            dr.b = f32_clamped(row.b);
            dr.tol = (float)row.tolerance;
            dr.out_row = r;
            dr.flags = flags;
        } else if (row.kind == 1) {
            if (nq >= SCG_MAX_QUADRATIC) return fail(SCG_ERR_INVALID, "scg_set_constraints: too many quadratic rows");
            QuadRow& q = ct->quad[nq++];
            const double* P = p_mats + (size_t)row.p_index * SCG_MAX_NV * SCG_MAX_NV;
            for (int a = 0; a < SCG_MAX_NV * SCG_MAX_NV; ++a) q.P[a] = (float)P[a];
            q.b = f32_clamped(row.b);
            q.tol = (float)row.tolerance;
            q.out_row = r;
            q.flags = flags;
        }
    }
    ct->n_dense = nd;
    ct->n_quad = nq;
    HIP_TRY(hipMemcpy(h->d_con, ct, sizeof(ConTable), hipMemcpyHostToDevice));
    h->n_con_rows = n_rows;
    h->kp.n_con_rows = n_rows;
    return SCG_OK;
}

int scg_reset(scg_handle* h, float* state, float* state_lo, int32_t* step_count, float* ep_return,
              int32_t* ep_length, float* obs, const uint8_t* mask, const float* reset_state, float* inertial,
              uint32_t* episode_id, uint32_t reset_serial, int64_t ld, int64_t n, void* stream) {
    if (!h || !state || !step_count) return fail(SCG_ERR_INVALID, "scg_reset: null argument");
    GUARD_DEVICE(h, "scg_reset");
    if (n < 0 || ld < n) return fail(SCG_ERR_INVALID, "scg_reset: need 0 <= n <= ld (n=%lld ld=%lld)", (long long)n, (long long)ld);
    if (n == 0) return SCG_OK;
    if (n > (int64_t)1 << 30) return fail(SCG_ERR_INVALID, "scg_reset: n > 2^30 envs per device (32-bit lane offsets)");
    if ((h->kp.flags & F_TRACKING) && h->kp.obs_goal_horizon > 0 && obs && !h->d_ref)
        return fail(SCG_ERR_INVALID, "scg_reset: traj_tracking needs scg_set_reference() first");
    if (h->dist_needs_episode && !episode_id)
        return fail(SCG_ERR_INVALID, "scg_reset: a per-episode step_offset needs the episode_id plane");
    if (h->dist_obs && obs && obs == state)
        return fail(SCG_ERR_INVALID, "scg_reset: an observation disturbance needs an obs buffer of its own (obs must not alias state)");
    if (int rc = join_async(h, (hipStream_t)stream, "scg_reset")) return rc;
    HIP_TRY(launch_reset(h->system, h->kp, h->d_tb, state, state_lo, step_count, ep_return, ep_length, obs, mask,
                         reset_state, inertial, episode_id, h->d_ref, reset_serial, ld, n, (hipStream_t)stream));
    return SCG_OK;
}

// Argument checks shared by every form of the step (scg_step, scg_step_async, the rollouts validate through it).
static int step_check(const scg_handle* h, const scg_step_io* io, int64_t n, const char* who) {
    if (!h || !io) return fail(SCG_ERR_INVALID, "%s: null argument", who);
    GUARD_DEVICE(h, who);
    if (io->struct_size != sizeof(scg_step_io))
        return fail(SCG_ERR_INVALID, "%s: struct_size %u != %zu", who, io->struct_size, sizeof(scg_step_io));
    if (n < 0 || io->ld < n) return fail(SCG_ERR_INVALID, "%s: need 0 <= n <= ld (n=%lld ld=%lld)", who, (long long)n, (long long)io->ld);
    if (!io->action || !io->state || !io->step_count || !io->reward || !io->done)
        return fail(SCG_ERR_INVALID, "%s: action, state, step_count, reward and done are mandatory", who);
    if ((h->kp.flags & F_TRACKING) && !h->d_ref)
        return fail(SCG_ERR_INVALID, "%s: traj_tracking needs scg_set_reference() first", who);
    if (h->kp.obs_goal_horizon > 0 && !io->obs)
        return fail(SCG_ERR_INVALID, "%s: obs_goal_horizon > 0 needs an obs buffer (obs does not alias state)", who);
    if (io->obs_noise && !io->obs)
        return fail(SCG_ERR_INVALID, "%s: obs_noise needs an obs buffer", who);
    if (io->terminal_record && io->terminal_stride != 0 && (io->terminal_stride < h->obs_dim + 2 || (io->terminal_stride & 3)))
        return fail(SCG_ERR_INVALID, "%s: terminal_stride must be 0 (planes [obs_dim + 2][ld]) or a multiple of 4 >= obs_dim + 2 = %d (got %lld)",
                    who, h->obs_dim + 2, (long long)io->terminal_stride);
    if (io->action_rows > 1 || (io->action_rows && (reinterpret_cast<uintptr_t>(io->action) % (4u * h->nu)) != 0))
        return fail(SCG_ERR_INVALID, "%s: row-major actions must be %d-byte aligned (action_rows in {0, 1})", who, 4 * h->nu);
    if (int rc = check_dist_io(h, io, who)) return rc;
    if (io->state_lo_kind > 2) return fail(SCG_ERR_INVALID, "%s: state_lo_kind must be 0, 1 or 2", who);
    if (io->state_lo && io->state_lo_kind == 2 && (reinterpret_cast<uintptr_t>(io->state_lo) & 1u))
        return fail(SCG_ERR_INVALID, "%s: int16 residual planes (state_lo_kind = 2) must be 2-byte aligned", who);
    if (io->state_lo && io->state_lo_kind >= 1) {
        if (h->system != SCG_QUAD_3D)
            return fail(SCG_ERR_UNSUPPORTED, "%s: attitude-only residual planes (state_lo_kind = 1, 2) are defined for the 3-D quadrotor", who);
        if (h->kp.dist || io->obs || io->inertial || io->action_disturbance || io->dynamics_force || io->stats || io->episode_id ||
            (h->kp.flags & (F_GND | F_DRAG | F_RAND_GENERAL)) || !(h->kp.flags & F_Q_DIAG))
            return fail(SCG_ERR_UNSUPPORTED, "%s: state_lo_kind = 1 / 2 supports the lean and info configurations only "
                                             "(no disturbances, aero terms, per-env inertia, full Q / R, obs buffer)", who);
    }
    if (n > (int64_t)1 << 30) return fail(SCG_ERR_INVALID, "%s: n > 2^30 envs per device (32-bit lane offsets)", who);
    return SCG_OK;
}

int scg_step(scg_handle* h, const scg_step_io* io, int64_t n, void* stream) {
    if (int rc = step_check(h, io, n, "scg_step")) return rc;
    if (n == 0) return SCG_OK;
    if (int rc = join_async(h, (hipStream_t)stream, "scg_step")) return rc;
    KParams kp = h->kp;
    if (h->cfg.lone_launch_forms >= 0) kp.flags |= F_LONE;      // a single dependent launch: the forms that are faster alone
    if (io->serial_word) { kp.serial_base = io->serial_word; kp.flags |= F_SERIAL_WORD; }
    HIP_TRY(launch_step(h->system, kp, *io, h->d_tb, h->d_con, h->d_ref, n, h->num_cus, (hipStream_t)stream));
    return SCG_OK;
}

int scg_step_async(scg_handle* h, const scg_step_io* io, int64_t n, void* stream) {
    if (int rc = step_check(h, io, n, "scg_step_async")) return rc;
    if (n == 0) return SCG_OK;
    if (int rc = ensure_aux_queue(h, "scg_step_async")) return rc;
    hipStream_t s = (hipStream_t)stream, q = h->aux_stream;
    if (s != q) {
        // the policy's writes (io->action) and whatever else the caller queued come first
        HIP_TRY(hipEventRecord(h->ev_async_in, s));
        HIP_TRY(hipStreamWaitEvent(q, h->ev_async_in, 0));
    }
    KParams kp = h->kp;
    if (h->cfg.lone_launch_forms >= 0) kp.flags |= F_LONE;
    if (io->serial_word) { kp.serial_base = io->serial_word; kp.flags |= F_SERIAL_WORD; }
    HIP_TRY(launch_step(h->system, kp, *io, h->d_tb, h->d_con, h->d_ref, n, h->num_cus, q));
    h->async_pending = true;
    return SCG_OK;
}

int scg_step_wait(scg_handle* h, void* stream) {
    if (!h) return fail(SCG_ERR_INVALID, "scg_step_wait: null handle");
    GUARD_DEVICE(h, "scg_step_wait");
    if (int rc = join_async(h, (hipStream_t)stream, "scg_step_wait")) return rc;
    h->async_pending = false;
    return SCG_OK;
}

int scg_step_queue(scg_handle* h, void** stream) {
    if (!h || !stream) return fail(SCG_ERR_INVALID, "scg_step_queue: null argument");
    GUARD_DEVICE(h, "scg_step_queue");
    if (int rc = ensure_aux_queue(h, "scg_step_queue")) return rc;
    *stream = (void*)h->aux_stream;
    return SCG_OK;
}

// The same buffers seen from env `first` on: every plane pointer advanced by `first` elements.
static scg_step_io shifted_io(const scg_step_io& io, int64_t first) {
    scg_step_io s = io;
    auto f32 = [&](const float* p) { return p ? p + first : p; };
    auto u8 = [&](uint8_t* p) { return p ? p + first : p; };
    s.action = f32(io.action);
    s.action_disturbance = f32(io.action_disturbance);
    s.dynamics_force = f32(io.dynamics_force);
    s.obs_noise = f32(io.obs_noise);
    s.reset_state = f32(io.reset_state);
    s.inertial = const_cast<float*>(f32(io.inertial));
    s.state = const_cast<float*>(f32(io.state));
    s.state_lo = const_cast<float*>(f32(io.state_lo));
    s.state_out = const_cast<float*>(f32(io.state_out));
    if (io.state_lo && io.state_lo_kind == 2)                            // int16 planes: `first` elements are 2 bytes each
        s.state_lo = reinterpret_cast<float*>(reinterpret_cast<int16_t*>(io.state_lo) + first);
    s.step_count = io.step_count ? io.step_count + first : nullptr;
    s.ep_return = const_cast<float*>(f32(io.ep_return));
    s.ep_length = io.ep_length ? io.ep_length + first : nullptr;
    s.episode_id = io.episode_id ? io.episode_id + first : nullptr;
    s.obs = const_cast<float*>(f32(io.obs));
    s.reward = const_cast<float*>(f32(io.reward));
    s.done = u8(io.done);
    s.constraint_values = const_cast<float*>(f32(io.constraint_values));
    s.constraint_violation = u8(io.constraint_violation);
    s.truncated = u8(io.truncated);
    s.goal_reached = u8(io.goal_reached);
    s.mse = const_cast<float*>(f32(io.mse));
    s.terminal_record = io.terminal_record ? io.terminal_record + first * (io.terminal_stride ? io.terminal_stride : 1) : nullptr;   // records | planes
    return s;
}

// ---- hipGraph replay of a small-batch rollout -----------------------------------------------------------------------
// Returns kGraphNotTaken when the caller should issue plain launches (first sighting of a signature, the caller's stream is
// itself being captured, capture unavailable), else the status of the replayed rollout.  The capture happens on the handle's
// own queue (the legacy default stream cannot be captured; nothing executes during a capture), the replay on the caller's
// stream; a failed capture is always ended (never left open), its graph destroyed, and the handle stays on plain launches
// from then on.
constexpr int kGraphNotTaken = 1;
constexpr int32_t kGraphMinSteps = 8;

static int rollout_by_graph(scg_handle* h, const scg_step_io* io, int64_t n, int32_t steps, int64_t action_stride,
                            int32_t action_period, hipStream_t s0) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s0, &cs) != hipSuccess) { (void)hipGetLastError(); return kGraphNotTaken; }
    if (cs != hipStreamCaptureStatusNone) return kGraphNotTaken;        // the caller captures this rollout itself: plain launches
    scg_handle::RolloutKey key;
    std::memset(&key, 0, sizeof(key));
    std::memcpy(&key.io, io, sizeof(scg_step_io));
    key.io.reset_serial = 0;
    key.n = n; key.action_stride = action_stride; key.steps = steps; key.action_period = action_period;
    scg_handle::GraphEntry* hit = nullptr;
    for (int q = 0; q < scg_handle::kGraphSlots; ++q)
        if (h->graphs[q].exec && std::memcmp(&h->graphs[q].key, &key, sizeof(key)) == 0) hit = &h->graphs[q];
    if (!hit) {
        const bool second = h->last_key_valid && std::memcmp(&h->last_key, &key, sizeof(key)) == 0;
        h->last_key = key;
        h->last_key_valid = true;
        if (!second) return kGraphNotTaken;                             // first sighting: plain launches, remember the signature
        if (int rc = scg_step(h, io, 0, s0)) return rc;                 // validates the buffers, launches nothing (n = 0)
        if (int rc = ensure_aux_queue(h, "scg_rollout")) return rc;
        // the slot first (an empty one, else the least recently used): its own serial word is baked into the nodes
        scg_handle::GraphEntry* slot = &h->graphs[0];
        for (int q = 0; q < scg_handle::kGraphSlots; ++q) {
            if (!h->graphs[q].exec) { slot = &h->graphs[q]; break; }
            if (h->graphs[q].last_use < slot->last_use) slot = &h->graphs[q];
        }
        if ((!slot->d_serial && hipMalloc(&slot->d_serial, sizeof(uint32_t)) != hipSuccess) ||
            (!slot->replayed && hipEventCreateWithFlags(&slot->replayed, hipEventDisableTiming) != hipSuccess)) {
            (void)hipGetLastError();
            h->graph_broken = true;
            return kGraphNotTaken;
        }
        hipStream_t s1 = h->aux_stream;
        if (hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            h->graph_broken = true;
            return kGraphNotTaken;
        }
        KParams kp = h->kp;
        if (h->cfg.lone_launch_forms > 0) kp.flags |= F_LONE;            // single-queue launches are dependent: each runs alone
        kp.serial_base = slot->d_serial;                                // the nodes read the serial of the rollout being replayed
        kp.flags |= F_SERIAL_WORD;
        scg_step_io local = *io;
        hipError_t err = hipSuccess;
        for (int32_t k = 0; k < steps && err == hipSuccess; ++k) {
            local.action = io->action + (int64_t)(k % action_period) * action_stride;
            local.reset_serial = (uint32_t)k;                           // offset from *d_serial
            err = launch_step(h->system, kp, local, h->d_tb, h->d_con, h->d_ref, n, h->num_cus, s1);
        }
        hipGraph_t graph = nullptr;
        const hipError_t end = hipStreamEndCapture(s1, &graph);         // ALWAYS: a capture is never left open
        hipGraphExec_t exec = nullptr;
        if (err == hipSuccess && end == hipSuccess && graph) err = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        else if (err == hipSuccess) err = end != hipSuccess ? end : hipErrorUnknown;
        if (graph) (void)hipGraphDestroy(graph);
        if (err != hipSuccess || !exec) {
            (void)hipGetLastError();
            h->graph_broken = true;                                     // plain launches from now on (this call included)
            return kGraphNotTaken;
        }
        retire_graph_entry(*slot);                                      // least recently used: waits for ITS last replay only
        slot->key = key;
        slot->exec = exec;
        hit = slot;
    }
    hit->last_use = ++h->graph_clock;
    // Replayed on the CALLER's stream, in order with its other work.  (Replaying on the handle's queue, forked from / joined into
    // the caller's stream by events, costs 1.1 us MORE per node than plain launches when that stream is the legacy default
    // stream -- tools/ubench/graph_launch.hip: 4.09 vs 3.20 plain vs 2.99 replayed in place.)  One signature must not be in flight
    // on two streams at once: its replays share the entry's serial word (distinct signatures have distinct words).
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)hit->d_serial, (int)io->reset_serial, 1, s0));
    HIP_TRY(hipGraphLaunch(hit->exec, s0));
    HIP_TRY(hipEventRecord(hit->replayed, s0));
    return SCG_OK;
}

// Step k of a rollout that writes into a trajectory sink: the same buffers with this step's output pointers.  When the
// observation IS the state (no obs buffer, or the shared goal-horizon buffer obs == state) the state ping-pongs through the
// sink: step k reads slab k - 1 (step 0: io.state) and writes slab k.
static bool sink_obs_is_state(const scg_step_io& io) { return !io.obs || io.obs == io.state; }

static scg_step_io sunk_io(const scg_step_io& io, const scg_trajectory_sink* sink, int32_t k) {
    scg_step_io s = io;
    if (!sink) return s;
    if (sink->obs) {
        float* slab = sink->obs + (int64_t)k * sink->obs_step_stride;
        if (sink_obs_is_state(io)) {
            s.state = k == 0 ? io.state : sink->obs + (int64_t)(k - 1) * sink->obs_step_stride;
            s.state_out = slab;
            if (io.obs) s.obs = slab;
        } else {
            s.obs = slab;
        }
    }
    if (sink->reward) s.reward = sink->reward + (int64_t)k * sink->reward_step_stride;
    if (sink->done) s.done = sink->done + (int64_t)k * sink->done_step_stride;
    if (sink->constraint_values) s.constraint_values = sink->constraint_values + (int64_t)k * sink->constraint_values_step_stride;
    if (sink->constraint_violation) s.constraint_violation = sink->constraint_violation + (int64_t)k * sink->constraint_violation_step_stride;
    if (sink->truncated) s.truncated = sink->truncated + (int64_t)k * sink->truncated_step_stride;
    if (sink->mse) s.mse = sink->mse + (int64_t)k * sink->mse_step_stride;
    return s;
}

// The per-step launches of scg_rollout / scg_rollout_trajectory (`sink` nullable).
static int rollout_launches(scg_handle* h, const scg_step_io* io, const scg_trajectory_sink* sink, int64_t n, int32_t steps,
                            int64_t action_stride, int32_t action_period, void* stream, const char* who) {
    // Two launch queues.  A step kernel spends ~5 us of its ~28 us (2^20 envs) ramping up and draining; the envs are
    // independent, so the batch is cut in two halves whose step chains run on two streams: the tail of one half's
    // kernel overlaps the body of the other's (measured 24.9 vs 29.0 us per step of the whole batch; four queues are
    // slower again).  Per env the arithmetic, the Philox counters and therefore the results are unchanged.
    const int queues = h->cfg.rollout_queues == 1 ? 1 : 2;
    // Default: batches of at least 2^19 envs.  At 2^18 (C3) the two queues are host-bound -- two launches of a ~4 us kernel per
    // step -- and land anywhere between 7.6 and 10.1 us per step by box and by run, where one queue replayed from the graph below
    // holds 7.9-8.0; up to 393216 envs the graph is level or ahead for the 2-D and the 3-D quadrotor; at 2^19 the two queues win
    // by 19 % (profiles/r04/c3_queues_ab.txt).
    const int64_t split_min = h->cfg.rollout_split_min > 0 ? h->cfg.rollout_split_min : (int64_t)262144;
    const int64_t half = (n / 2) / 256 * 256;
    if (queues == 2 && steps >= 2 && half >= split_min && half > 0) {
        if (int rc = ensure_aux_queue(h, who)) return rc;
        hipStream_t s0 = (hipStream_t)stream, s1 = h->aux_stream;
        HIP_TRY(hipEventRecord(h->ev_fork, s0));                            // nothing runs on s1 yet: plain returns are safe
        HIP_TRY(hipStreamWaitEvent(s1, h->ev_fork, 0));
        KParams ka = h->kp;
        if (io->serial_word) { ka.serial_base = io->serial_word; ka.flags |= F_SERIAL_WORD; }
        KParams kb = ka;
        kb.env_base += half;                                               // Philox streams follow the global env index
        // From here on s1 may hold work on the caller's buffers: whatever fails, s1 is joined back into s0 before this
        // call returns, so the caller never sees an un-joined queue running against its tensors.
        hipError_t err = hipSuccess;
        const char* what = "";
        for (int32_t k = 0; k < steps && err == hipSuccess; ++k) {
            scg_step_io a = sunk_io(*io, sink, k);
            a.action = io->action + (int64_t)(k % action_period) * action_stride;
            a.reset_serial = io->reset_serial + (uint32_t)k;
            const scg_step_io b = shifted_io(a, half);
            err = launch_step(h->system, ka, a, h->d_tb, h->d_con, h->d_ref, half, h->num_cus, s0);
            what = "launch_step (first half)";
            if (err != hipSuccess) break;
            err = launch_step(h->system, kb, b, h->d_tb, h->d_con, h->d_ref, n - half, h->num_cus, s1);
            what = "launch_step (second half)";
        }
        hipError_t je = hipEventRecord(h->ev_join, s1);
        if (je == hipSuccess) je = hipStreamWaitEvent(s0, h->ev_join, 0);
        if (je != hipSuccess) {
            // the device-side join could not be queued: fall back to a host-side one so that s1 is quiescent on return
            (void)hipStreamSynchronize(s1);
            if (err == hipSuccess) { err = je; what = "join of the second launch queue"; }
        }
        if (err != hipSuccess) return fail(SCG_ERR_HIP, "%s: %s failed: %s", who, what, hipGetErrorString(err));
        return SCG_OK;
    }
    // Small batches: a step kernel lasts a few microseconds and the launches themselves set the pace.  A rollout signature
    // (buffers, n, steps, action layout) seen twice in a row is captured into a hipGraph once and replayed from then on
    // (measured at the BASELINE sizes: CartPole 2^16 4.96 -> 4.30, Quadrotor-2D 2^18 8.48 -> 7.85 us per step).  Not with a
    // sink: its launch-free form is the fused one.
    if (!sink && !io->serial_word && n > 0 && steps >= kGraphMinSteps && h->cfg.rollout_graph >= 0 && !h->graph_broken) {
        const int rc = rollout_by_graph(h, io, n, steps, action_stride, action_period, (hipStream_t)stream);
        if (rc != kGraphNotTaken) return rc;
    }
    KParams kp1 = h->kp;
    if (h->cfg.lone_launch_forms > 0) kp1.flags |= F_LONE;               // single-queue launches are dependent: each runs alone
    if (io->serial_word) { kp1.serial_base = io->serial_word; kp1.flags |= F_SERIAL_WORD; }
    for (int32_t k = 0; k < steps; ++k) {
        scg_step_io local = sunk_io(*io, sink, k);
        local.action = io->action + (int64_t)(k % action_period) * action_stride;
        local.reset_serial = io->reset_serial + (uint32_t)k;
        HIP_TRY(launch_step(h->system, kp1, local, h->d_tb, h->d_con, h->d_ref, n, h->num_cus, (hipStream_t)stream));
    }
    return SCG_OK;
}

static int rollout_check(const scg_handle* h, const scg_step_io* io, int64_t n, int32_t steps, int64_t action_stride,
                         int32_t action_period, const char* who) {
    if (!h || !io) return fail(SCG_ERR_INVALID, "%s: null argument", who);
    GUARD_DEVICE(h, who);
    if (steps < 0 || action_period < 1 || action_stride < 0)
        return fail(SCG_ERR_INVALID, "%s: need steps >= 0, action_period >= 1, action_stride >= 0", who);
    if (io->struct_size == sizeof(scg_step_io) && io->action_rows)
        return fail(SCG_ERR_INVALID, "%s: actions are [period][nu][ld] planes (action_rows must be 0)", who);
    if (io->struct_size == sizeof(scg_step_io) && io->state_out && io->state_out != io->state)
        return fail(SCG_ERR_INVALID, "%s: a multi-step rollout updates the state in place (state_out must be NULL)", who);
    return step_check(h, io, n, who);
}

int scg_rollout(scg_handle* h, const scg_step_io* io, int64_t n, int32_t steps, int64_t action_stride,
                int32_t action_period, void* stream) {
    if (int rc = rollout_check(h, io, n, steps, action_stride, action_period, "scg_rollout")) return rc;
    if (steps == 0 || n == 0) return SCG_OK;
    if (int rc = join_async(h, (hipStream_t)stream, "scg_rollout")) return rc;
    return rollout_launches(h, io, nullptr, n, steps, action_stride, action_period, stream, "scg_rollout");
}

int scg_rollout_trajectory(scg_handle* h, const scg_step_io* io, const scg_trajectory_sink* sink, int64_t n, int32_t steps,
                           int64_t action_stride, int32_t action_period, int32_t fused, void* stream) {
    const char* who = "scg_rollout_trajectory";
    if (!sink) return fail(SCG_ERR_INVALID, "%s: null sink", who);
    if (sink->struct_size != sizeof(scg_trajectory_sink))
        return fail(SCG_ERR_INVALID, "%s: sink struct_size %u != %zu", who, sink->struct_size, sizeof(scg_trajectory_sink));
    if (int rc = rollout_check(h, io, n, steps, action_stride, action_period, who)) return rc;
    if (sink->flags & ~SCG_SINK_KEEP_STATE_IN_SINK) return fail(SCG_ERR_INVALID, "%s: unknown sink flags 0x%x", who, sink->flags);
    const int64_t ld = io->ld;
    const int obs_planes = io->obs ? h->obs_dim : h->nx;
    struct Member { const void* p; int64_t stride, least; const char* name; };
    const Member members[7] = {{sink->obs, sink->obs_step_stride, (int64_t)obs_planes * ld, "obs"},
                               {sink->reward, sink->reward_step_stride, n, "reward"},
                               {sink->done, sink->done_step_stride, n, "done"},
                               {sink->constraint_values, sink->constraint_values_step_stride, (int64_t)h->n_con_rows * ld, "constraint_values"},
                               {sink->constraint_violation, sink->constraint_violation_step_stride, n, "constraint_violation"},
                               {sink->truncated, sink->truncated_step_stride, n, "truncated"},
                               {sink->mse, sink->mse_step_stride, n, "mse"}};
    for (const Member& m : members)
        if (m.p && steps > 1 && m.stride < m.least)
            return fail(SCG_ERR_INVALID, "%s: %s_step_stride %lld is smaller than one step's slab (%lld elements)", who, m.name,
                        (long long)m.stride, (long long)m.least);
    if ((sink->constraint_values || sink->constraint_violation) && h->n_con_rows == 0)
        return fail(SCG_ERR_INVALID, "%s: a constraint sink needs scg_set_constraints() first", who);
    if (sink->obs && sink_obs_is_state(*io) && (sink->obs == io->state || (reinterpret_cast<uintptr_t>(sink->obs) & 3u)))
        return fail(SCG_ERR_INVALID, "%s: the obs sink must be a buffer of its own (the state ping-pongs through it)", who);
    if (steps == 0 || n == 0) return SCG_OK;
    if (int rc = join_async(h, (hipStream_t)stream, who)) return rc;
    if (fused && fused_rollout_ok(h->kp, *io) && !sink->truncated && !sink->mse) {
        RolloutArgs ro{steps, action_period, action_stride, sink->obs, sink->obs_step_stride, sink->reward, sink->reward_step_stride,
                       sink->done, sink->done_step_stride, sink->constraint_values, sink->constraint_values_step_stride,
                       sink->constraint_violation, sink->constraint_violation_step_stride};
        KParams kp = h->kp;
        if (io->serial_word) { kp.serial_base = io->serial_word; kp.flags |= F_SERIAL_WORD; }
        HIP_TRY(launch_rollout_fused(h->system, kp, *io, h->d_tb, h->d_con, h->d_ref, n, ro, h->num_cus, (hipStream_t)stream));
        return SCG_OK;
    }
    if (int rc = rollout_launches(h, io, sink, n, steps, action_stride, action_period, stream, who)) return rc;
    if (sink->obs && sink_obs_is_state(*io) && !(sink->flags & SCG_SINK_KEEP_STATE_IN_SINK)) {
        // the chain of states ran through the sink: bring the last slab home (once per call, behind the last step and the join)
        const float* last = sink->obs + (int64_t)(steps - 1) * sink->obs_step_stride;
        const size_t bytes = ((size_t)(obs_planes - 1) * (size_t)ld + (size_t)n) * sizeof(float);
        HIP_TRY(hipMemcpyAsync(io->state, last, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return SCG_OK;
}

int scg_rollout_fused(scg_handle* h, const scg_step_io* io, int64_t n, int32_t steps, int64_t action_stride,
                      int32_t action_period, void* stream) {
    if (int rc = rollout_check(h, io, n, steps, action_stride, action_period, "scg_rollout_fused")) return rc;
    if (steps == 0 || n == 0) return SCG_OK;
    if (!fused_rollout_ok(h->kp, *io)) return scg_rollout(h, io, n, steps, action_stride, action_period, stream);
    if (int rc = join_async(h, (hipStream_t)stream, "scg_rollout_fused")) return rc;
    RolloutArgs ro;
    std::memset(&ro, 0, sizeof(ro));
    ro.steps = steps; ro.period = action_period; ro.stride = action_stride;
    KParams kp = h->kp;
    if (io->serial_word) { kp.serial_base = io->serial_word; kp.flags |= F_SERIAL_WORD; }
    HIP_TRY(launch_rollout_fused(h->system, kp, *io, h->d_tb, h->d_con, h->d_ref, n, ro, h->num_cus, (hipStream_t)stream));
    return SCG_OK;
}

int scg_eval_constraints(scg_handle* h, const float* state, const float* action, float* c_val,
                         uint8_t* violation, int64_t ld, int64_t n, void* stream) {
    if (!h || !state) return fail(SCG_ERR_INVALID, "scg_eval_constraints: null argument");
    GUARD_DEVICE(h, "scg_eval_constraints");
    if (n < 0 || ld < n) return fail(SCG_ERR_INVALID, "scg_eval_constraints: need 0 <= n <= ld");
    if (n == 0 || h->n_con_rows == 0) return SCG_OK;
    if (n > (int64_t)1 << 30) return fail(SCG_ERR_INVALID, "scg_eval_constraints: n > 2^30 envs per device");
    HIP_TRY(launch_eval_constraints(h->system, h->d_con, state, action, c_val, violation, ld, n, (hipStream_t)stream));
    return SCG_OK;
}

int scg_gae(const float* rew, const float* val, const float* mask, const uint8_t* done, const float* term_val,
            const float* last_val, float* ret, float* adv, float* rew_out, double* stats, int32_t T, int64_t n, int64_t ld,
            float gamma, float gae_lambda, int32_t use_gae, void* stream) {
    if (!rew || !val || !last_val || !ret || !adv) return fail(SCG_ERR_INVALID, "scg_gae: null argument");
    if ((mask == nullptr) == (done == nullptr)) return fail(SCG_ERR_INVALID, "scg_gae: exactly one of mask (float planes) and done (byte planes) must be given");
    if (T < 1 || n < 0 || ld < n) return fail(SCG_ERR_INVALID, "scg_gae: need T >= 1 and 0 <= n <= ld");
    if (n == 0) return SCG_OK;
    if (n > (int64_t)1 << 30) return fail(SCG_ERR_INVALID, "scg_gae: n > 2^30");
    HIP_TRY(launch_gae(rew, val, mask, done, term_val, last_val, ret, adv, rew_out, stats, T, n, ld, gamma, gae_lambda,
                       use_gae, (hipStream_t)stream));
    return SCG_OK;
}

int scg_normalize_advantages(float* adv, const double* stats, int32_t T, int64_t n, int64_t ld, float eps,
                             void* stream) {
    if (!adv || !stats) return fail(SCG_ERR_INVALID, "scg_normalize_advantages: null argument");
    if (T < 1 || n < 0 || ld < n) return fail(SCG_ERR_INVALID, "scg_normalize_advantages: need T >= 1 and 0 <= n <= ld");
    if (n == 0) return SCG_OK;
    HIP_TRY(launch_normalize(adv, stats, T, n, ld, eps, (hipStream_t)stream));
    return SCG_OK;
}

int scg_copy_segments(const scg_copy_segment* segments, int32_t count, void* stream) {
    if (!segments) return fail(SCG_ERR_INVALID, "scg_copy_segments: null argument");
    if (count < 1 || count > SCG_MAX_COPY_SEGMENTS)
        return fail(SCG_ERR_INVALID, "scg_copy_segments: count %d is not in 1 .. %d", (int)count, SCG_MAX_COPY_SEGMENTS);
    for (int j = 0; j < count; ++j) {
        const scg_copy_segment& g = segments[j];
        if (g.rows < 0 || g.cols < 0 || g.rows > ((int64_t)1 << 40) || g.cols > ((int64_t)1 << 40) ||
            (g.cols > 0 && g.rows > ((int64_t)1 << 40) / g.cols))
            return fail(SCG_ERR_INVALID, "scg_copy_segments: segment %d: rows / cols out of range", j);
        if (g.rows * g.cols > 0 && (!g.src || !g.dst)) return fail(SCG_ERR_INVALID, "scg_copy_segments: segment %d: null pointer", j);
        if (g.src_row_stride < 0 || g.src_col_stride < 0 || g.dst_row_stride < 0 || g.dst_col_stride < 0)
            return fail(SCG_ERR_INVALID, "scg_copy_segments: segment %d: negative stride", j);
        // a destination whose elements alias (a zero stride over more than one element) has no defined result
        if ((g.rows > 1 && g.dst_row_stride == 0) || (g.cols > 1 && g.dst_col_stride == 0))
            return fail(SCG_ERR_INVALID, "scg_copy_segments: segment %d: destination stride 0", j);
    }
    HIP_TRY(launch_copy_segments(segments, count, (hipStream_t)stream));
    return SCG_OK;
}

int scg_model_eval(scg_handle* h, const float* state, const float* input, const float* inertial, float* xdot,
                   float* xnext, int64_t ld, int64_t n, void* stream) {
    if (!h) return fail(SCG_ERR_INVALID, "scg_model_eval: null handle");
    GUARD_DEVICE(h, "scg_model_eval");
    if (!state || !input || (!xdot && !xnext)) return fail(SCG_ERR_INVALID, "scg_model_eval: null argument");
    if (n < 0 || ld < n) return fail(SCG_ERR_INVALID, "scg_model_eval: need 0 <= n <= ld");
    if (n > (int64_t)1 << 30) return fail(SCG_ERR_INVALID, "scg_model_eval: n > 2^30");
    if (n == 0) return SCG_OK;
    HIP_TRY(launch_model_eval(h->system, h->kp, h->d_tb, state, input, inertial, xdot, xnext, ld, n, (hipStream_t)stream));
    return SCG_OK;
}

int scg_model_flow(scg_handle* h, const double* state, const double* input, const double* inertial, double dt, double rtol,
                   double atol, int32_t max_doublings, double* xf, double* dxf_dx0, double* dxf_dp, int32_t* intervals,
                   int64_t ld, int64_t n, void* stream) {
    if (!h) return fail(SCG_ERR_INVALID, "scg_model_flow: null handle");
    GUARD_DEVICE(h, "scg_model_flow");
    if (!state || !input || (!xf && !dxf_dx0 && !dxf_dp)) return fail(SCG_ERR_INVALID, "scg_model_flow: null argument");
    if (n < 0 || ld < n) return fail(SCG_ERR_INVALID, "scg_model_flow: need 0 <= n <= ld");
    const bool sens = dxf_dx0 || dxf_dp;
    if (n > ((int64_t)1 << (sens ? 26 : 30))) return fail(SCG_ERR_INVALID, "scg_model_flow: n > 2^%d", sens ? 26 : 30);
    if (!(dt > 0.0) || !std::isfinite(dt)) return fail(SCG_ERR_INVALID, "scg_model_flow: dt must be positive and finite");
    if (!(rtol >= 0.0) || !(atol >= 0.0) || !(rtol + atol > 0.0) || !std::isfinite(rtol + atol))
        return fail(SCG_ERR_INVALID, "scg_model_flow: need rtol >= 0, atol >= 0, not both zero");
    if (max_doublings < 0 || max_doublings > 16) return fail(SCG_ERR_INVALID, "scg_model_flow: max_doublings must be in 0..16");
    if (n == 0) return SCG_OK;
    HIP_TRY(launch_model_flow(h->system, h->phd, state, input, inertial, dt, rtol, atol, max_doublings, xf, dxf_dx0, dxf_dp,
                              intervals, ld, n, (hipStream_t)stream));
    return SCG_OK;
}

int scg_linearize(scg_handle* h, const float* state, const float* input, const float* inertial, float* dfdx,
                  float* dfdu, double disc_dt, int64_t ld, int64_t n, void* stream) {
    if (!h) return fail(SCG_ERR_INVALID, "scg_linearize: null handle");
    GUARD_DEVICE(h, "scg_linearize");
    if (!state || !input || !dfdx || !dfdu) return fail(SCG_ERR_INVALID, "scg_linearize: null argument");
    if (n < 0 || ld < n) return fail(SCG_ERR_INVALID, "scg_linearize: need 0 <= n <= ld");
    if (n > (int64_t)1 << 30) return fail(SCG_ERR_INVALID, "scg_linearize: n > 2^30");
    if (!(disc_dt >= 0.0)) return fail(SCG_ERR_INVALID, "scg_linearize: disc_dt must be >= 0");
    HIP_TRY(launch_linearize(h->system, h->kp.ph, state, input, inertial, dfdx, dfdu, (float)disc_dt, ld, n,
                             (hipStream_t)stream));
    return SCG_OK;
}

int scg_fc_linear(scg_handle* h, const float* x_eval, const float* u_eval, const float* state, const float* input,
                  const float* inertial, float* out, int64_t ld, int64_t n, void* stream) {
    if (!h) return fail(SCG_ERR_INVALID, "scg_fc_linear: null handle");
    GUARD_DEVICE(h, "scg_fc_linear");
    if (!u_eval || !state || !input || !out) return fail(SCG_ERR_INVALID, "scg_fc_linear: null argument");
    if (n < 0 || ld < n) return fail(SCG_ERR_INVALID, "scg_fc_linear: need 0 <= n <= ld");
    if (n > (int64_t)1 << 30) return fail(SCG_ERR_INVALID, "scg_fc_linear: n > 2^30");
    HIP_TRY(launch_fc_linear(h->system, h->kp.ph, x_eval, u_eval, state, input, inertial, out, ld, n, (hipStream_t)stream));
    return SCG_OK;
}

int scg_discretize_exact(const float* A, const float* B, int32_t nx, int32_t m, double dt, float* Ad, float* Bd,
                         const float* z0, const float* w, float* z1, int64_t n, void* stream) {
    if (!A || !B) return fail(SCG_ERR_INVALID, "scg_discretize_exact: null argument");
    if (nx < 1 || m < 1 || nx + m > 16) return fail(SCG_ERR_INVALID, "scg_discretize_exact: need nx >= 1, m >= 1, nx + m <= 16 (got %d + %d)", nx, m);
    if (!Ad && !Bd && !z1) return fail(SCG_ERR_INVALID, "scg_discretize_exact: no output requested");
    if (z1 && !z0) return fail(SCG_ERR_INVALID, "scg_discretize_exact: z1 needs z0");
    if (!(dt > 0.0) || !std::isfinite(dt)) return fail(SCG_ERR_INVALID, "scg_discretize_exact: dt must be positive and finite");
    if (n < 0 || n > (int64_t)1 << 30) return fail(SCG_ERR_INVALID, "scg_discretize_exact: need 0 <= n <= 2^30");
    if (n == 0) return SCG_OK;
    // CU count of the current device, asked once per device (this entry point is a per-control-step call of LQR / MPC loops)
    static int cu_cache[64] = {0};
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        cus = __atomic_load_n(&cu_cache[dev], __ATOMIC_RELAXED);
        if (cus == 0 && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            __atomic_store_n(&cu_cache[dev], cus, __ATOMIC_RELAXED);
    }
    HIP_TRY(launch_discretize_exact(A, B, nx, m, (float)dt, Ad, Bd, z0, w, z1, n, cus, (hipStream_t)stream));
    return SCG_OK;
}

int scg_quadratic_loss(scg_handle* h, const double* Q, const double* R, const float* state, const float* input,
                       const float* xr, int64_t xr_ld, const float* ur, int64_t ur_ld, float* l, float* l_x,
                       float* l_u, int64_t ld, int64_t n, void* stream) {
    if (!h) return fail(SCG_ERR_INVALID, "scg_quadratic_loss: null handle");
    GUARD_DEVICE(h, "scg_quadratic_loss");
    if (!Q || !R || !state || !input || !xr || !ur) return fail(SCG_ERR_INVALID, "scg_quadratic_loss: null argument");
    if (n < 0 || ld < n) return fail(SCG_ERR_INVALID, "scg_quadratic_loss: need 0 <= n <= ld");
    if ((xr_ld != 0 && xr_ld < n) || (ur_ld != 0 && ur_ld < n))
        return fail(SCG_ERR_INVALID, "scg_quadratic_loss: reference plane stride smaller than n");
    if (n > (int64_t)1 << 30) return fail(SCG_ERR_INVALID, "scg_quadratic_loss: n > 2^30");
    HIP_TRY(launch_loss(h->system, Q, R, state, input, xr, xr_ld, ur, ur_ld, l, l_x, l_u, ld, n, (hipStream_t)stream));
    return SCG_OK;
}

int scg_mean_std_normalize(const float* x, float* y, int32_t d, int64_t ld_x, int64_t ld_y, int64_t n,
                           const double* state_in, double* state_out, double* moments, float clip, double epsilon,
                           void* stream) {
    if (!x || !state_in) return fail(SCG_ERR_INVALID, "scg_mean_std_normalize: null argument");
    if (!y && !state_out) return fail(SCG_ERR_INVALID, "scg_mean_std_normalize: y may be null only in an updating call (moments only)");
    if (state_out && !moments) return fail(SCG_ERR_INVALID, "scg_mean_std_normalize: an updating call needs the moments scratch");
    if (state_out == state_in) return fail(SCG_ERR_INVALID, "scg_mean_std_normalize: state_out must not alias state_in");
    if (d < 1 || d > 65535) return fail(SCG_ERR_INVALID, "scg_mean_std_normalize: need 1 <= d <= 65535");
    if (n < 0 || ld_x < n || (y && ld_y < n)) return fail(SCG_ERR_INVALID, "scg_mean_std_normalize: need 0 <= n <= ld_x, ld_y");
    if (n > (int64_t)1 << 30) return fail(SCG_ERR_INVALID, "scg_mean_std_normalize: n > 2^30");
    if (!(clip >= 0.f) || !(epsilon >= 0.0)) return fail(SCG_ERR_INVALID, "scg_mean_std_normalize: clip and epsilon must be >= 0");
    if (n == 0) return SCG_OK;                     // np.mean of an empty batch is NaN in the reference; refuse to poison the tracker
    HIP_TRY(launch_mean_std_normalize(x, y, d, ld_x, ld_y, n, state_in, state_out, moments, clip, epsilon,
                                      (hipStream_t)stream));
    return SCG_OK;
}

int scg_reward_normalize(const float* rew, const uint8_t* done, float* ret, float* out, int64_t n,
                         const double* state_in, double* state_out, double* moments, float gamma, float clip,
                         double epsilon, int32_t legacy_index_reset, void* stream) {
    if (!rew || !out || !state_in) return fail(SCG_ERR_INVALID, "scg_reward_normalize: null argument");
    if (state_out && (!moments || !done || !ret))
        return fail(SCG_ERR_INVALID, "scg_reward_normalize: an updating call needs done, ret and the moments scratch");
    if (state_out == state_in) return fail(SCG_ERR_INVALID, "scg_reward_normalize: state_out must not alias state_in");
    if (n < 0 || n > (int64_t)1 << 30) return fail(SCG_ERR_INVALID, "scg_reward_normalize: need 0 <= n <= 2^30");
    if (!(clip >= 0.f) || !(epsilon >= 0.0)) return fail(SCG_ERR_INVALID, "scg_reward_normalize: clip and epsilon must be >= 0");
    if (n == 0) return SCG_OK;
    HIP_TRY(launch_reward_normalize(rew, done, ret, out, n, state_in, state_out, moments, gamma, clip, epsilon,
                                    legacy_index_reset, (hipStream_t)stream));
    return SCG_OK;
}

int scg_mean_std_moments(const float* x, int32_t d, int64_t ld_x, int64_t n, const double* state_in, double* scratch, double* sums,
                         void* stream) {
    if (!x || !state_in || !scratch || !sums) return fail(SCG_ERR_INVALID, "scg_mean_std_moments: null argument");
    if (d < 1 || d > 65535) return fail(SCG_ERR_INVALID, "scg_mean_std_moments: need 1 <= d <= 65535");
    if (n < 1 || ld_x < n || n > (int64_t)1 << 30) return fail(SCG_ERR_INVALID, "scg_mean_std_moments: need 1 <= n <= ld_x, n <= 2^30");
    HIP_TRY(launch_mean_std_moments(x, d, ld_x, n, state_in, scratch, sums, (hipStream_t)stream));
    return SCG_OK;
}

int scg_mean_std_normalize_sums(const float* x, float* y, int32_t d, int64_t ld_x, int64_t ld_y, int64_t n, const double* state_in,
                                double* state_out, const double* sums, double batch_count, float clip, double epsilon, void* stream) {
    if (!x || !state_in || !state_out || !sums) return fail(SCG_ERR_INVALID, "scg_mean_std_normalize_sums: null argument");
    if (state_out == state_in) return fail(SCG_ERR_INVALID, "scg_mean_std_normalize_sums: state_out must not alias state_in");
    if (d < 1 || d > 65535) return fail(SCG_ERR_INVALID, "scg_mean_std_normalize_sums: need 1 <= d <= 65535");
    if (n < 0 || ld_x < n || (y && ld_y < n) || n > (int64_t)1 << 30)
        return fail(SCG_ERR_INVALID, "scg_mean_std_normalize_sums: need 0 <= n <= ld_x, ld_y and n <= 2^30");
    if (!(batch_count >= 1.0) || batch_count < (double)n)
        return fail(SCG_ERR_INVALID, "scg_mean_std_normalize_sums: batch_count (the rows of all shards) must be >= max(n, 1)");
    if (!(clip >= 0.f) || !(epsilon >= 0.0)) return fail(SCG_ERR_INVALID, "scg_mean_std_normalize_sums: clip and epsilon must be >= 0");
    HIP_TRY(launch_mean_std_apply_sums(x, n > 0 ? y : nullptr, d, ld_x, ld_y, n, state_in, state_out, sums, batch_count, clip, epsilon,
                                       (hipStream_t)stream));
    return SCG_OK;
}

int scg_reward_moments(const float* rew, const uint8_t* done, float* ret, int64_t n, const double* state_in, double* scratch,
                       double* sums, float gamma, void* stream) {
    if (!rew || !done || !ret || !state_in || !scratch || !sums) return fail(SCG_ERR_INVALID, "scg_reward_moments: null argument");
    if (n < 1 || n > (int64_t)1 << 30) return fail(SCG_ERR_INVALID, "scg_reward_moments: need 1 <= n <= 2^30");
    HIP_TRY(launch_reward_moments(rew, done, ret, n, state_in, scratch, sums, gamma, (hipStream_t)stream));
    return SCG_OK;
}

int scg_reward_normalize_sums(const float* rew, float* out, int64_t n, const double* state_in, double* state_out, const double* sums,
                              double batch_count, float clip, double epsilon, void* stream) {
    if (!rew || !out || !state_in || !state_out || !sums) return fail(SCG_ERR_INVALID, "scg_reward_normalize_sums: null argument");
    if (state_out == state_in) return fail(SCG_ERR_INVALID, "scg_reward_normalize_sums: state_out must not alias state_in");
    if (n < 1 || n > (int64_t)1 << 30) return fail(SCG_ERR_INVALID, "scg_reward_normalize_sums: need 1 <= n <= 2^30");
    if (!(batch_count >= (double)n)) return fail(SCG_ERR_INVALID, "scg_reward_normalize_sums: batch_count (the envs of all shards) must be >= n");
    if (!(clip >= 0.f) || !(epsilon >= 0.0)) return fail(SCG_ERR_INVALID, "scg_reward_normalize_sums: clip and epsilon must be >= 0");
    HIP_TRY(launch_reward_apply_sums(rew, out, n, state_in, state_out, sums, batch_count, clip, epsilon, (hipStream_t)stream));
    return SCG_OK;
}

int scg_num_constraints(const scg_handle* h) { return h ? h->n_con_rows : 0; }

int scg_obs_dim(const scg_handle* h) { return h ? h->obs_dim : 0; }

int scg_rollout_graph_info(const scg_handle* h, int32_t* cached, int64_t* replays) {
    if (!h) return fail(SCG_ERR_INVALID, "scg_rollout_graph_info: null handle");
    int c = 0;
    for (int q = 0; q < scg_handle::kGraphSlots; ++q) c += h->graphs[q].exec ? 1 : 0;
    if (cached) *cached = h->graph_broken ? -1 : c;
    if (replays) *replays = (int64_t)h->graph_clock;
    return SCG_OK;
}

int scg_kernel_info(const scg_handle* h, int32_t* block, int32_t* grid_for_n, int32_t* lds_bytes, int64_t n) {
    if (!h) return fail(SCG_ERR_INVALID, "scg_kernel_info: null handle");
    LaunchGeom g = step_geometry(h->kp, h->system, n, h->num_cus);
    if (block) *block = g.block;
    if (grid_for_n) *grid_for_n = g.grid;
    if (lds_bytes) *lds_bytes = g.lds_bytes;
    return SCG_OK;
}

}  // extern "C"
