// Driver of the CPU sanitizer build of the C-ABI validation layer (scg_capi.hip compiled as host C++ against the
// stand-in HIP runtime of this directory, -fsanitize=address,undefined).  Exercises every argument check, the table
// compilers (reference / constraints / disturbances) at their limits, and the two-queue rollout's error paths.
// Exit code 0 = every expectation held and the sanitizers reported nothing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/scg_rollout.h"
#include "hip_host_shim.h"

static int g_fail = 0;
#define EXPECT(cond)                                                                                        \
    do {                                                                                                    \
        if (!(cond)) { std::fprintf(stderr, "EXPECT failed at %s:%d: %s   [%s]\n", __FILE__, __LINE__, #cond, scg_last_error()); ++g_fail; } \
    } while (0)

static scg_config base_config(int system) {
    scg_config c;
    std::memset(&c, 0, sizeof(c));
    c.struct_size = sizeof(c);
    c.abi_version = SCG_ABI_VERSION;
    c.system = system;
    c.integrator = SCG_RK4;
    c.substeps = 1;
    c.task = SCG_STABILIZATION;
    c.cost = SCG_RL_REWARD;
    c.ctrl_steps = 300;
    c.dt = 1.0 / 60;
    c.mass = 0.03454; c.gravity = 9.8; c.arm = 0.0397; c.ixx = 1.4e-5; c.iyy = 1.4e-5; c.izz = 2.17e-5; c.kf = 3.16e-10; c.km = 7.94e-12;
    c.pwm2rpm_scale = 0.2685; c.pwm2rpm_const = 4070.3; c.pwm_min = 20000; c.pwm_max = 65535;
    c.pole_length = 0.5; c.pole_mass = 0.1; c.cart_mass = 1.0;
    c.motor_map = 1;
    for (int j = 0; j < SCG_MAX_NU; ++j) { c.act_low[j] = 0.028; c.act_high[j] = 0.148; c.u_goal[j] = 0.0846; c.rew_act_weight[j] = 1e-4; }
    for (int j = 0; j < SCG_MAX_NX; ++j) { c.rew_state_weight[j] = 1.0; c.oob_low[j] = (j & 1) ? -3.5e38 : -5.0; c.oob_high[j] = (j & 1) ? 3.5e38 : 5.0; c.init_rand_low[j] = -0.1; c.init_rand_high[j] = 0.1; }
    for (int j = 0; j < SCG_MAX_NX; ++j) c.Q[j * SCG_MAX_NX + j] = 1.0;
    for (int j = 0; j < SCG_MAX_NU; ++j) c.R[j * SCG_MAX_NU + j] = 1.0;
    c.goal_tolerance = 0.05;
    c.done_on_out_of_bound = 1;
    c.randomized_init = 1;
    c.auto_reset = 1;
    c.seed = 1337;
    return c;
}

static int dims_nx(int s) { return s == 1 ? 2 : s == 2 ? 6 : s == 3 ? 12 : 4; }
static int dims_nu(int s) { return s == 1 ? 1 : s == 2 ? 2 : s == 3 ? 4 : 1; }

// a step_io whose pointers are non-null host buffers (never dereferenced by the layer: it only validates and forwards)
struct Buffers {
    std::vector<float> f;
    std::vector<int32_t> i;
    std::vector<uint8_t> b;
    scg_step_io io;
    explicit Buffers(int64_t ld) : f(64), i(64), b(64) {
        std::memset(&io, 0, sizeof(io));
        io.struct_size = sizeof(io);
        io.ld = ld;
        io.action = f.data(); io.state = f.data(); io.step_count = i.data(); io.reward = f.data(); io.done = b.data();
    }
};

static void check_create_validation() {
    scg_handle* h = nullptr;
    scg_config c = base_config(SCG_QUAD_3D);
    EXPECT(scg_create(nullptr, &h) == SCG_ERR_INVALID);
    EXPECT(scg_create(&c, nullptr) == SCG_ERR_INVALID);
    { scg_config d = c; d.struct_size = 8; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.abi_version = SCG_ABI_VERSION - 1; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.system = 9; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.integrator = 5; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.task = 2; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.cost = 2; EXPECT(scg_create(&d, &h) == SCG_ERR_UNSUPPORTED); }
    { scg_config d = c; d.substeps = 0; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.substeps = 65; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.dt = 0.0; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.ctrl_steps = 0; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.obs_goal_horizon = SCG_MAX_GOAL_HORIZON + 1; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.mass = 0.0; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = base_config(SCG_CARTPOLE); d.pole_mass = -1.0; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.rollout_queues = 3; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.rollout_split_min = -1; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.grid_cap_per_cu = -2; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.init_rand_kind[3] = 99; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.inertial_rand_kind[1] = -1; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.randomized_inertial = 1; d.inertial_rand_low[0] = -1.0; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    // choice() carries its number of values in *_rand_low: 1, 2 or 3
    for (double cnt : {0.0, 4.0, 1.5, -1.0}) {
        scg_config d = c; d.init_rand_kind[2] = SCG_RAND_CHOICE; d.init_rand_low[2] = cnt; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID);
        scg_config e = c; e.randomized_inertial = 1; e.inertial_rand_kind[1] = SCG_RAND_CHOICE; e.inertial_rand_low[1] = cnt;
        e.inertial_rand_p[1][0] = e.inertial_rand_p[1][1] = e.inertial_rand_p[1][2] = 1e-6; EXPECT(scg_create(&e, &h) == SCG_ERR_INVALID);
    }
    for (double cnt : {1.0, 2.0, 3.0}) {
        scg_config d = c; d.init_rand_kind[2] = SCG_RAND_CHOICE; d.init_rand_low[2] = cnt;
        scg_handle* ok = nullptr; EXPECT(scg_create(&d, &ok) == SCG_OK); EXPECT(scg_destroy(ok) == SCG_OK);
    }
    // parameter ranges of the rejection samplers and the weighted choice
    { scg_config d = c; d.init_rand_kind[0] = SCG_RAND_GAMMA; d.init_rand_p[0][0] = 0.0; d.init_rand_p[0][1] = 1.0; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.init_rand_kind[0] = SCG_RAND_GAMMA; d.init_rand_p[0][0] = 2.0; d.init_rand_p[0][1] = -1.0; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.init_rand_kind[1] = SCG_RAND_BETA; d.init_rand_p[1][0] = 2.0; d.init_rand_p[1][1] = 0.0; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.init_rand_kind[2] = SCG_RAND_CHOICE_P; d.init_rand_low[2] = 0.6; d.init_rand_high[2] = 0.5; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.init_rand_kind[2] = SCG_RAND_CHOICE_P; d.init_rand_low[2] = 0.6; d.init_rand_high[2] = 1.5; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.init_rand_kind[0] = SCG_RAND_GAMMA; d.init_rand_p[0][0] = 0.5; d.init_rand_p[0][1] = 0.1;
      d.init_rand_kind[1] = SCG_RAND_BETA; d.init_rand_p[1][0] = 2.0; d.init_rand_p[1][1] = 3.0;
      d.init_rand_kind[2] = SCG_RAND_CHOICE_P; d.init_rand_low[2] = 0.25; d.init_rand_high[2] = 0.75;
      scg_handle* ok = nullptr; EXPECT(scg_create(&d, &ok) == SCG_OK); EXPECT(scg_destroy(ok) == SCG_OK); }
    // a weighted choice of an inertial property must not reach a non-positive value with a non-zero probability
    { scg_config d = c; d.randomized_inertial = 1; d.inertial_rand_kind[0] = SCG_RAND_CHOICE_P; d.inertial_rand_low[0] = 0.5; d.inertial_rand_high[0] = 1.0;
      d.inertial_rand_p[0][0] = 0.0; d.inertial_rand_p[0][1] = -1.0; d.inertial_rand_p[0][2] = -1.0; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID);
      d.inertial_rand_high[0] = 0.5; d.inertial_rand_low[0] = 0.5; d.inertial_rand_p[0][2] = 0.001;        // p1 has probability 0 now
      scg_handle* ok = nullptr; EXPECT(scg_create(&d, &ok) == SCG_OK); EXPECT(scg_destroy(ok) == SCG_OK); }
    EXPECT(g_shim.live_allocations == 0);             // no failed create leaks its tables
    EXPECT(scg_destroy(nullptr) == SCG_OK);
    EXPECT(scg_version() == SCG_ABI_VERSION && scg_sizeof_config() == (int)sizeof(scg_config) && scg_sizeof_step_io() == (int)sizeof(scg_step_io));
}

static void check_tables(int system) {
    const int nx = dims_nx(system), nu = dims_nu(system), nv = nx + nu;
    scg_config c = base_config(system);
    c.task = SCG_TRAJ_TRACKING;
    c.cost = SCG_QUADRATIC;
    c.Q[1] = 0.25;                                    // non-diagonal Q: the full-matrix path of the table builder
    scg_handle* h = nullptr;
    EXPECT(scg_create(&c, &h) == SCG_OK && h != nullptr);
    if (!h) return;
    EXPECT(scg_obs_dim(h) == nx);
    // reference table: too short for the quadratic cost, then exact sizes incl. one whose byte count is not a multiple of 16
    std::vector<double> table((size_t)301 * nx, 0.5);
    EXPECT(scg_set_reference(h, nullptr, 300) == SCG_ERR_INVALID);
    EXPECT(scg_set_reference(h, table.data(), 0) == SCG_ERR_INVALID);
    EXPECT(scg_set_reference(h, table.data(), 299) == SCG_ERR_INVALID);
    EXPECT(scg_set_reference(h, table.data(), 301) == SCG_OK);
    EXPECT(scg_set_reference(h, table.data(), 300) == SCG_OK);
    // constraints: the maximum number of rows of every kind the tables can hold
    std::vector<scg_constraint_row> rows(SCG_MAX_CONSTRAINT_ROWS + 1);
    std::vector<double> pm((size_t)SCG_MAX_QUADRATIC * SCG_MAX_NV * SCG_MAX_NV, 0.0);
    for (int q = 0; q < SCG_MAX_QUADRATIC; ++q) for (int a = 0; a < nv; ++a) pm[(size_t)q * SCG_MAX_NV * SCG_MAX_NV + a * SCG_MAX_NV + a] = 1.0 + q;
    for (size_t r = 0; r < rows.size(); ++r) {
        std::memset(&rows[r], 0, sizeof(rows[r]));
        rows[r].b = 1.0 + 0.01 * (double)r;
        if (r % 3 == 0) { rows[r].coef[r % nv] = (r & 1) ? 1.0 : -1.0; rows[r].strict = 1; }            // bound row
        else { rows[r].coef[r % nv] = 0.5; rows[r].coef[(r + 1) % nv] = -2.0; rows[r].has_tolerance = 1; rows[r].tolerance = 0.1; }   // dense row
    }
    for (int q = 0; q < SCG_MAX_QUADRATIC; ++q) { rows[1 + 3 * q].kind = 1; rows[1 + 3 * q].p_index = q; }
    EXPECT(scg_set_constraints(h, rows.data(), SCG_MAX_CONSTRAINT_ROWS + 1, pm.data(), SCG_MAX_QUADRATIC) == SCG_ERR_INVALID);
    EXPECT(scg_set_constraints(h, nullptr, 3, pm.data(), 0) == SCG_ERR_INVALID);
    EXPECT(scg_set_constraints(h, rows.data(), SCG_MAX_CONSTRAINT_ROWS, pm.data(), SCG_MAX_QUADRATIC + 1) == SCG_ERR_INVALID);
    EXPECT(scg_set_constraints(h, rows.data(), SCG_MAX_CONSTRAINT_ROWS, pm.data(), SCG_MAX_QUADRATIC) == SCG_OK);
    EXPECT(scg_num_constraints(h) == SCG_MAX_CONSTRAINT_ROWS);
    { auto bad = rows; bad[1].p_index = SCG_MAX_QUADRATIC; EXPECT(scg_set_constraints(h, bad.data(), 8, pm.data(), SCG_MAX_QUADRATIC) == SCG_ERR_INVALID); }
    { auto bad = rows; bad[1].kind = 7; EXPECT(scg_set_constraints(h, bad.data(), 8, pm.data(), SCG_MAX_QUADRATIC) == SCG_ERR_INVALID); }
    if (nv < SCG_MAX_NV) { auto bad = rows; bad[0].coef[SCG_MAX_NV - 1] = 1.0; EXPECT(scg_set_constraints(h, bad.data(), 8, pm.data(), SCG_MAX_QUADRATIC) == SCG_ERR_INVALID); }
    EXPECT(scg_set_constraints(h, rows.data(), 0, nullptr, 0) == SCG_OK);        // clearing
    // disturbances: SCG_MAX_DISTURBANCES per mode, one too many, malformed impulse / random offset
    std::vector<scg_disturbance> ds;
    for (int mode = 0; mode < 3; ++mode)
        for (int e = 0; e < SCG_MAX_DISTURBANCES; ++e) {
            scg_disturbance d;
            std::memset(&d, 0, sizeof(d));
            d.mode = mode; d.kind = (mode + e) % 5;
            d.step_offset = e == 0 ? -1 : 3; d.max_step = 300; d.duration = 2; d.decay_rate = 0.5; d.magnitude = 0.1; d.scale = 0.2; d.frequency = 1.5;
            d.has_mask = e & 1;
            for (int j = 0; j < SCG_MAX_NX; ++j) { d.low[j] = -0.1; d.high[j] = 0.1; d.mask[j] = j & 1; }
            ds.push_back(d);
        }
    EXPECT(scg_set_disturbances(h, ds.data(), (int32_t)ds.size()) == SCG_OK);
    { auto more = ds; more.push_back(ds[0]); EXPECT(scg_set_disturbances(h, more.data(), (int32_t)more.size()) == SCG_ERR_INVALID); }
    { auto bad = ds; bad[0].mode = 3; EXPECT(scg_set_disturbances(h, bad.data(), 1) == SCG_ERR_INVALID); }
    { auto bad = ds; bad[0].kind = 9; EXPECT(scg_set_disturbances(h, bad.data(), 1) == SCG_ERR_INVALID); }
    { auto bad = ds; bad[0].kind = SCG_DIST_IMPULSE; bad[0].duration = 0; EXPECT(scg_set_disturbances(h, bad.data(), 1) == SCG_ERR_INVALID); }
    { auto bad = ds; bad[0].kind = SCG_DIST_STEP; bad[0].step_offset = -1; bad[0].max_step = 0; EXPECT(scg_set_disturbances(h, bad.data(), 1) == SCG_ERR_INVALID); }
    EXPECT(scg_set_disturbances(h, nullptr, 2) == SCG_ERR_INVALID);
    // with a per-episode offset and an observation entry attached, step() asks for the planes they need
    { Buffers b(128); EXPECT(scg_step(h, &b.io, 100, nullptr) == SCG_ERR_INVALID); }
    // an observation entry needs an obs buffer of its own: obs == state (the goal-horizon sharing) is refused
    { Buffers b(128); std::vector<uint32_t> ep(128); b.io.episode_id = ep.data(); b.io.obs = (float*)b.io.state;
      EXPECT(scg_step(h, &b.io, 100, nullptr) == SCG_ERR_INVALID); }
    EXPECT(scg_set_disturbances(h, nullptr, 0) == SCG_OK);
    EXPECT(scg_destroy(h) == SCG_OK);
    EXPECT(g_shim.live_allocations == 0);
}

static void check_step_and_rollout() {
    scg_config c = base_config(SCG_QUAD_3D);
    scg_handle* h = nullptr;
    EXPECT(scg_create(&c, &h) == SCG_OK);
    Buffers b(1 << 20);
    EXPECT(scg_step(nullptr, &b.io, 1, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_step(h, nullptr, 1, nullptr) == SCG_ERR_INVALID);
    { Buffers x(64); x.io.struct_size = 12; EXPECT(scg_step(h, &x.io, 1, nullptr) == SCG_ERR_INVALID); }
    { Buffers x(64); EXPECT(scg_step(h, &x.io, 65, nullptr) == SCG_ERR_INVALID); EXPECT(scg_step(h, &x.io, -1, nullptr) == SCG_ERR_INVALID); }
    { Buffers x(64); x.io.reward = nullptr; EXPECT(scg_step(h, &x.io, 8, nullptr) == SCG_ERR_INVALID); }
    { Buffers x(64); x.io.obs_noise = x.f.data(); EXPECT(scg_step(h, &x.io, 8, nullptr) == SCG_ERR_INVALID); }
    { Buffers x(64); x.io.obs = (float*)x.io.state; x.io.obs_noise = x.f.data(); EXPECT(scg_step(h, &x.io, 8, nullptr) == SCG_ERR_INVALID); }
    { Buffers x(64); x.io.obs = (float*)x.io.state; EXPECT(scg_step(h, &x.io, 8, nullptr) == SCG_OK); }
    // state_out: a ping-pong step is fine, obs may share the buffer the state is written to (not the one it is read from),
    // the multi-step rollouts update in place only
    { Buffers x(64); std::vector<float> o(64); x.io.state_out = o.data(); EXPECT(scg_step(h, &x.io, 8, nullptr) == SCG_OK);
      x.io.obs = o.data(); EXPECT(scg_step(h, &x.io, 8, nullptr) == SCG_OK);
      x.io.obs = (float*)x.io.state; EXPECT(scg_step(h, &x.io, 8, nullptr) == SCG_ERR_INVALID);
      x.io.obs = nullptr; EXPECT(scg_rollout(h, &x.io, 8, 2, 0, 1, nullptr) == SCG_ERR_INVALID);
      EXPECT(scg_rollout_fused(h, &x.io, 8, 2, 0, 1, nullptr) == SCG_ERR_INVALID);
      x.io.state_out = x.io.state; EXPECT(scg_rollout(h, &x.io, 8, 2, 0, 1, nullptr) == SCG_OK); }
    { Buffers x(64); x.io.terminal_record = x.f.data(); x.io.terminal_stride = 13; EXPECT(scg_step(h, &x.io, 8, nullptr) == SCG_ERR_INVALID); }
    { Buffers x(64); x.io.terminal_record = x.f.data(); x.io.terminal_stride = 16; EXPECT(scg_step(h, &x.io, 8, nullptr) == SCG_OK); }
    { Buffers x(64); x.io.terminal_record = x.f.data(); x.io.terminal_stride = 0; EXPECT(scg_step(h, &x.io, 8, nullptr) == SCG_OK); }   // SoA planes [obs_dim + 2][ld]
    { Buffers x(64); x.io.terminal_record = x.f.data(); x.io.terminal_stride = 8; EXPECT(scg_step(h, &x.io, 8, nullptr) == SCG_ERR_INVALID); }   // a record shorter than obs_dim + 2
    { Buffers x(64); x.io.action_rows = 2; EXPECT(scg_step(h, &x.io, 8, nullptr) == SCG_ERR_INVALID); }
    { Buffers x(64); x.io.action_rows = 1; x.io.action = x.f.data() + 1; EXPECT(scg_step(h, &x.io, 8, nullptr) == SCG_ERR_INVALID); }
    { Buffers x((int64_t)1 << 31); EXPECT(scg_step(h, &x.io, ((int64_t)1 << 30) + 1, nullptr) == SCG_ERR_INVALID); }
    const int64_t before = g_shim.launches;
    EXPECT(scg_step(h, &b.io, 0, nullptr) == SCG_OK && g_shim.launches == before);      // n = 0 launches nothing
    EXPECT(scg_step(h, &b.io, 1000, nullptr) == SCG_OK && g_shim.launches == before + 1);
    // rollout argument checks
    EXPECT(scg_rollout(h, &b.io, 1000, -1, 0, 1, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_rollout(h, &b.io, 1000, 3, 0, 0, nullptr) == SCG_ERR_INVALID);
    { Buffers x(64); x.io.action_rows = 1; EXPECT(scg_rollout(h, &x.io, 8, 3, 0, 1, nullptr) == SCG_ERR_INVALID); }
    EXPECT(scg_rollout(h, &b.io, 1000, 0, 0, 1, nullptr) == SCG_OK);
    // small batch: one queue
    shim_reset();
    EXPECT(scg_rollout(h, &b.io, 1000, 5, 16, 2, nullptr) == SCG_OK);
    EXPECT(g_shim.step_launches_main == 5 && g_shim.step_launches_aux == 0);
    // large batch: two queues, forked and joined by events, second half keyed by its global env index
    shim_reset();
    const int64_t n = 700001, half = (n / 2) / 256 * 256;
    EXPECT(scg_rollout(h, &b.io, n, 7, 16, 2, nullptr) == SCG_OK);
    EXPECT(g_shim.step_launches_main == 7 && g_shim.step_launches_aux == 7);
    EXPECT(g_shim.aux_streams_open == 0 && g_shim.event_records == 2 && g_shim.stream_waits == 2);
    EXPECT(g_shim.last_n == n - half && g_shim.last_env_base == half);
    // a launch failing anywhere between the fork and the join: the error comes back AND the second queue is joined
    for (int k = 0; k < 9; ++k) {
        shim_reset();
        g_shim.fail_launch_after = k;
        EXPECT(scg_rollout(h, &b.io, n, 7, 16, 2, nullptr) == SCG_ERR_HIP);
        EXPECT(g_shim.aux_streams_open == 0);
        EXPECT(g_shim.step_launches_main + g_shim.step_launches_aux == k);
    }
    // the fork event cannot be recorded: nothing was queued on the second stream, plain error
    shim_reset();
    g_shim.fail_event_record_after = 0;
    EXPECT(scg_rollout(h, &b.io, n, 7, 16, 2, nullptr) == SCG_ERR_HIP && g_shim.aux_streams_open == 0 && g_shim.launches == 0);
    // the join event cannot be recorded: the layer falls back to a host-side join of the second queue
    shim_reset();
    g_shim.fail_event_record_after = 1;
    EXPECT(scg_rollout(h, &b.io, n, 7, 16, 2, nullptr) == SCG_ERR_HIP && g_shim.aux_streams_open == 0 && g_shim.stream_syncs == 1);
    // residual-plane kinds: 0 all states (float), 1 / 2 attitude-only (float / int16): 3-D quadrotor only, kind <= 2, int16 planes 2-byte aligned
    {
        Buffers r(n);
        r.io.state_lo = r.f.data();
        r.io.state_lo_kind = 3;
        EXPECT(scg_step(h, &r.io, 8, nullptr) == SCG_ERR_INVALID);
        r.io.state_lo_kind = 2;
        EXPECT(scg_step(h, &r.io, 8, nullptr) == SCG_OK);                       // this handle is the 3-D quadrotor
        r.io.state_lo = reinterpret_cast<float*>(reinterpret_cast<char*>(r.f.data()) + 1);
        EXPECT(scg_step(h, &r.io, 8, nullptr) == SCG_ERR_INVALID);
    }
    // a caller's serial word (its own HIP graph of the step loop): forwarded to the kernel parameters; scg_rollout's internal graph
    // replay stands back
    {
        Buffers w(4096);
        uint32_t word = 0;
        w.io.serial_word = &word;
        shim_reset();
        EXPECT(scg_step(h, &w.io, 4096, nullptr) == SCG_OK && g_shim.last_serial_base == &word);
        EXPECT(scg_rollout(h, &w.io, 4096, 16, 16, 2, nullptr) == SCG_OK && scg_rollout(h, &w.io, 4096, 16, 16, 2, nullptr) == SCG_OK);
        EXPECT(g_shim.graph_launches == 0 && g_shim.captured_nodes == 0 && g_shim.last_serial_base == &word);
        w.io.serial_word = nullptr;
        EXPECT(scg_step(h, &w.io, 4096, nullptr) == SCG_OK && g_shim.last_serial_base == nullptr);
    }
    shim_reset();
    EXPECT(scg_rollout_fused(h, &b.io, 1000, 4, 16, 2, nullptr) == SCG_OK && g_shim.launches == 1);
    EXPECT(scg_destroy(h) == SCG_OK);
    // per-handle launch policy: one queue only / a lower split threshold
    c.rollout_queues = 1;
    EXPECT(scg_create(&c, &h) == SCG_OK);
    shim_reset();
    EXPECT(scg_rollout(h, &b.io, n, 3, 16, 2, nullptr) == SCG_OK && g_shim.step_launches_aux == 0 && g_shim.step_launches_main == 3);
    EXPECT(scg_destroy(h) == SCG_OK);
    c.rollout_queues = 2; c.rollout_split_min = 256; c.grid_cap_per_cu = 2;
    EXPECT(scg_create(&c, &h) == SCG_OK);
    shim_reset();
    EXPECT(scg_rollout(h, &b.io, 1000, 3, 16, 2, nullptr) == SCG_OK && g_shim.step_launches_aux == 3 && g_shim.aux_streams_open == 0);
    int32_t blk = 0, grid = 0, lds = 0;
    EXPECT(scg_kernel_info(h, &blk, &grid, &lds, (int64_t)1 << 20) == SCG_OK && grid == 512);
    EXPECT(scg_destroy(h) == SCG_OK);
}

// hipGraph replay of small-batch rollouts: first sighting plain, second captured + replayed, later ones replayed; every way a
// capture can fail ends the capture, destroys what was built and falls back to plain launches.
static void check_rollout_graphs() {
    scg_config c = base_config(SCG_QUAD_2D);
    scg_handle* h = nullptr;
    const int64_t n = 4096;
    Buffers b(n), b2(n);
    EXPECT(scg_create(&c, &h) == SCG_OK);
    shim_reset();
    b.io.reset_serial = 100;
    EXPECT(scg_rollout(h, &b.io, n, 16, 16, 2, nullptr) == SCG_OK);                       // first sighting
    EXPECT(g_shim.step_launches_main == 16 && g_shim.graph_launches == 0 && g_shim.captured_nodes == 0);
    b.io.reset_serial = 116;
    EXPECT(scg_rollout(h, &b.io, n, 16, 16, 2, nullptr) == SCG_OK);                       // second: captured, replayed
    EXPECT(g_shim.step_launches_main == 16 && g_shim.captured_nodes == 16 && g_shim.graph_launches == 1 && g_shim.graph_nodes_replayed == 16);
    EXPECT(g_shim.captures_open == 0 && g_shim.graphs_live == 0 && g_shim.execs_live == 1 && g_shim.aux_streams_open == 0);
    EXPECT(g_shim.last_serial == 116 && g_shim.serial_memsets == 1);
    b.io.reset_serial = 132;
    EXPECT(scg_rollout(h, &b.io, n, 16, 16, 2, nullptr) == SCG_OK);                       // third: replay only
    EXPECT(g_shim.captured_nodes == 16 && g_shim.graph_launches == 2 && g_shim.last_serial == 132 && g_shim.aux_streams_open == 0);
    // a different signature (steps) starts over; the first one is still cached
    EXPECT(scg_rollout(h, &b.io, n, 12, 16, 2, nullptr) == SCG_OK && g_shim.graph_launches == 2);
    EXPECT(scg_rollout(h, &b.io, n, 12, 16, 2, nullptr) == SCG_OK && g_shim.graph_launches == 3 && g_shim.execs_live == 2);
    EXPECT(scg_rollout(h, &b.io, n, 16, 16, 2, nullptr) == SCG_OK && g_shim.graph_launches == 4 && g_shim.execs_live == 2);
    // at most 4 signatures are kept
    for (int steps = 20; steps < 26; ++steps) {
        EXPECT(scg_rollout(h, &b.io, n, steps, 16, 2, nullptr) == SCG_OK);
        EXPECT(scg_rollout(h, &b.io, n, steps, 16, 2, nullptr) == SCG_OK);
    }
    EXPECT(g_shim.execs_live == 4 && g_shim.captures_open == 0 && g_shim.graphs_live == 0);
    // another buffer set is another signature
    const int64_t before = g_shim.graph_launches;
    EXPECT(scg_rollout(h, &b2.io, n, 25, 16, 2, nullptr) == SCG_OK && g_shim.graph_launches == before);
    // short rollouts and a caller that is capturing itself are never graphed
    EXPECT(scg_rollout(h, &b.io, n, 7, 16, 2, nullptr) == SCG_OK && scg_rollout(h, &b.io, n, 7, 16, 2, nullptr) == SCG_OK && g_shim.graph_launches == before);
    g_shim.caller_stream_capturing = 1;
    EXPECT(scg_rollout(h, &b.io, n, 25, 16, 2, nullptr) == SCG_OK && g_shim.graph_launches == before);
    g_shim.caller_stream_capturing = 0;
    // whatever re-points the handle's tables drops the graphs
    EXPECT(scg_set_constraints(h, nullptr, 0, nullptr, 0) == SCG_OK && g_shim.execs_live == 0);
    EXPECT(scg_destroy(h) == SCG_OK);

    // capture failures: (a) a launch fails inside the capture, (b) instantiation fails, (c) the capture cannot begin
    for (int mode = 0; mode < 3; ++mode) {
        EXPECT(scg_create(&c, &h) == SCG_OK);
        shim_reset();
        EXPECT(scg_rollout(h, &b.io, n, 16, 16, 2, nullptr) == SCG_OK && g_shim.step_launches_main == 16);
        if (mode == 0) g_shim.fail_launch_after = 16 + 5;                                 // the 6th captured launch fails
        if (mode == 1) g_shim.fail_instantiate = 1;
        if (mode == 2) g_shim.fail_begin_capture = 1;
        const int rc = scg_rollout(h, &b.io, n, 16, 16, 2, nullptr);
        EXPECT(g_shim.captures_open == 0 && g_shim.graphs_live == 0 && g_shim.execs_live == 0 && g_shim.aux_streams_open == 0);
        EXPECT(g_shim.graph_launches == 0);
        if (mode == 0) EXPECT(rc == SCG_ERR_HIP);                                         // the fallback's plain launches hit the same injected failure
        else EXPECT(rc == SCG_OK && g_shim.step_launches_main == 32);                     // fallback: plain launches
        g_shim.fail_launch_after = -1; g_shim.fail_instantiate = 0; g_shim.fail_begin_capture = 0;
        const int64_t plain = g_shim.step_launches_main;
        EXPECT(scg_rollout(h, &b.io, n, 16, 16, 2, nullptr) == SCG_OK && scg_rollout(h, &b.io, n, 16, 16, 2, nullptr) == SCG_OK);
        EXPECT(g_shim.step_launches_main == plain + 32 && g_shim.graph_launches == 0);    // the handle stays on plain launches
        EXPECT(scg_destroy(h) == SCG_OK);
    }
    // switched off in the configuration
    c.rollout_graph = -1;
    EXPECT(scg_create(&c, &h) == SCG_OK);
    shim_reset();
    EXPECT(scg_rollout(h, &b.io, n, 16, 16, 2, nullptr) == SCG_OK && scg_rollout(h, &b.io, n, 16, 16, 2, nullptr) == SCG_OK);
    EXPECT(g_shim.step_launches_main == 32 && g_shim.graph_launches == 0 && g_shim.captured_nodes == 0);
    EXPECT(scg_destroy(h) == SCG_OK);
    { scg_config d = c; d.rollout_graph = 2; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.lone_launch_forms = 2; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    { scg_config d = c; d.lone_launch_forms = -2; EXPECT(scg_create(&d, &h) == SCG_ERR_INVALID); }
    EXPECT(g_shim.execs_live == 0 && g_shim.graphs_live == 0);
}

// scg_step_async / scg_step_wait (the VecEnv.step_async / step_wait split): the step goes to the handle's own queue behind an
// event of the caller's stream; the wait joins it back; every other launching entry point joins a pending step first.
static void check_async_step() {
    scg_config c = base_config(SCG_QUAD_3D);
    scg_handle* h = nullptr;
    EXPECT(scg_create(&c, &h) == SCG_OK);
    Buffers b(4096);
    EXPECT(scg_step_async(nullptr, &b.io, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_step_async(h, nullptr, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_step_wait(nullptr, nullptr) == SCG_ERR_INVALID);
    { Buffers x(64); x.io.reward = nullptr; EXPECT(scg_step_async(h, &x.io, 8, nullptr) == SCG_ERR_INVALID); }
    shim_reset();
    EXPECT(scg_step_wait(h, nullptr) == SCG_OK && g_shim.event_records == 0 && g_shim.stream_waits == 0);   // nothing pending: a no-op
    EXPECT(scg_step_async(h, &b.io, 0, nullptr) == SCG_OK && g_shim.launches == 0);                          // n = 0 launches nothing
    EXPECT(scg_step_async(h, &b.io, 4096, nullptr) == SCG_OK);
    EXPECT(g_shim.step_launches_aux == 1 && g_shim.step_launches_main == 0 && g_shim.event_records == 1 && g_shim.stream_waits == 1);
    EXPECT(g_shim.aux_streams_open == 1);                                                                    // pending on the handle's queue
    EXPECT(scg_step_wait(h, nullptr) == SCG_OK && g_shim.aux_streams_open == 0 && g_shim.event_records == 2 && g_shim.stream_waits == 2);
    EXPECT(scg_step_wait(h, nullptr) == SCG_OK && g_shim.event_records == 2);                                // joined already
    // a synchronous step (or a reset, a rollout) behind a pending asynchronous one joins it first
    EXPECT(scg_step_async(h, &b.io, 4096, nullptr) == SCG_OK && g_shim.aux_streams_open == 1);
    EXPECT(scg_step(h, &b.io, 4096, nullptr) == SCG_OK && g_shim.aux_streams_open == 0 && g_shim.step_launches_main == 1);
    EXPECT(scg_rollout(h, &b.io, 4096, 3, 16, 2, nullptr) == SCG_OK && g_shim.step_launches_main == 4);
    EXPECT(scg_step_wait(h, nullptr) == SCG_OK);
    // on the handle's own queue: no event at all
    void* q = nullptr;
    EXPECT(scg_step_queue(h, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_step_queue(h, &q) == SCG_OK && q != nullptr);
    shim_reset();
    EXPECT(scg_step_async(h, &b.io, 4096, q) == SCG_OK && g_shim.event_records == 0 && g_shim.stream_waits == 0 && g_shim.step_launches_aux == 1);
    EXPECT(scg_step_wait(h, q) == SCG_OK && g_shim.event_records == 0);
    // ... and a wait from ANOTHER stream after a step issued on the queue itself still joins
    EXPECT(scg_step_async(h, &b.io, 4096, q) == SCG_OK);
    EXPECT(scg_step_wait(h, nullptr) == SCG_OK && g_shim.event_records == 1 && g_shim.stream_waits == 1 && g_shim.aux_streams_open == 0);
    // the event of the caller's stream cannot be recorded: nothing was launched
    shim_reset();
    g_shim.fail_event_record_after = 0;
    EXPECT(scg_step_async(h, &b.io, 4096, nullptr) == SCG_ERR_HIP && g_shim.launches == 0 && g_shim.aux_streams_open == 0);
    shim_reset();
    EXPECT(scg_destroy(h) == SCG_OK);                                                                        // synchronises its queue first
}

// One handle per device: every handle-taking entry point refuses to run with another device current.
static void check_device_guard() {
    scg_config c = base_config(SCG_QUAD_2D);
    scg_handle* h = nullptr;
    EXPECT(scg_create(&c, &h) == SCG_OK);                                  // created with device 0 current
    Buffers b(64);
    std::vector<float> f(64);
    std::vector<double> d(64);
    scg_trajectory_sink sink;
    std::memset(&sink, 0, sizeof(sink));
    sink.struct_size = sizeof(sink);
    void* q = nullptr;
    const int64_t before = g_shim.launches;
    g_shim.current_device = 3;
    EXPECT(scg_step(h, &b.io, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(std::strstr(scg_last_error(), "device 0") && std::strstr(scg_last_error(), "device 3"));           // both indices are named
    EXPECT(scg_step_async(h, &b.io, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_step_wait(h, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_step_queue(h, &q) == SCG_ERR_INVALID);
    EXPECT(scg_rollout(h, &b.io, 8, 2, 0, 1, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_rollout_fused(h, &b.io, 8, 2, 0, 1, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_rollout_trajectory(h, &b.io, &sink, 8, 2, 0, 1, 0, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_reset(h, f.data(), nullptr, (int32_t*)f.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 8, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_set_reference(h, d.data(), 1) == SCG_ERR_INVALID);
    EXPECT(scg_set_constraints(h, nullptr, 0, nullptr, 0) == SCG_ERR_INVALID);
    EXPECT(scg_set_disturbances(h, nullptr, 0) == SCG_ERR_INVALID);
    EXPECT(scg_eval_constraints(h, f.data(), nullptr, f.data(), nullptr, 8, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_model_eval(h, f.data(), f.data(), nullptr, f.data(), nullptr, 8, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_model_flow(h, d.data(), d.data(), nullptr, 0.02, 1e-6, 1e-8, 8, d.data(), nullptr, nullptr, nullptr, 8, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_linearize(h, f.data(), f.data(), nullptr, f.data(), f.data(), 0.02, 8, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_fc_linear(h, nullptr, f.data(), f.data(), f.data(), nullptr, f.data(), 8, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_quadratic_loss(h, d.data(), d.data(), f.data(), f.data(), f.data(), 0, f.data(), 0, f.data(), nullptr, nullptr, 8, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(g_shim.launches == before);                                     // nothing was launched from the wrong device
    g_shim.current_device = 0;
    EXPECT(scg_step(h, &b.io, 8, nullptr) == SCG_OK);
    // a handle created with device 3 current belongs to device 3
    EXPECT(scg_destroy(h) == SCG_OK);
    g_shim.current_device = 3;
    EXPECT(scg_create(&c, &h) == SCG_OK && scg_step(h, &b.io, 8, nullptr) == SCG_OK);
    g_shim.current_device = 0;
    EXPECT(scg_step(h, &b.io, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_destroy(h) == SCG_OK);                                      // destroying is allowed from anywhere
}

// scg_rollout_trajectory: per-step output pointers, the state chain through the sink, the copy home, the fused form.
static void check_trajectory_sink() {
    scg_config c = base_config(SCG_QUAD_3D);
    scg_handle* h = nullptr;
    EXPECT(scg_create(&c, &h) == SCG_OK);
    const int64_t n = 1000, ld = 1024, T = 5;
    std::vector<float> state((size_t)12 * ld), sobs((size_t)T * 12 * ld), srew((size_t)T * ld);
    std::vector<uint8_t> sdone((size_t)T * ld);
    Buffers b(ld);
    b.io.state = state.data();
    scg_trajectory_sink sink;
    std::memset(&sink, 0, sizeof(sink));
    sink.struct_size = sizeof(sink);
    sink.obs = sobs.data(); sink.obs_step_stride = 12 * ld;
    sink.reward = srew.data(); sink.reward_step_stride = ld;
    sink.done = sdone.data(); sink.done_step_stride = ld;
    EXPECT(scg_sizeof_trajectory_sink() == (int)sizeof(scg_trajectory_sink));
    EXPECT(scg_rollout_trajectory(h, &b.io, nullptr, n, (int32_t)T, 16, 2, 0, nullptr) == SCG_ERR_INVALID);
    { scg_trajectory_sink bad = sink; bad.struct_size = 8; EXPECT(scg_rollout_trajectory(h, &b.io, &bad, n, (int32_t)T, 16, 2, 0, nullptr) == SCG_ERR_INVALID); }
    { scg_trajectory_sink bad = sink; bad.flags = 6; EXPECT(scg_rollout_trajectory(h, &b.io, &bad, n, (int32_t)T, 16, 2, 0, nullptr) == SCG_ERR_INVALID); }
    { scg_trajectory_sink bad = sink; bad.obs_step_stride = 12 * ld - 1; EXPECT(scg_rollout_trajectory(h, &b.io, &bad, n, (int32_t)T, 16, 2, 0, nullptr) == SCG_ERR_INVALID); }
    { scg_trajectory_sink bad = sink; bad.reward_step_stride = n - 1; EXPECT(scg_rollout_trajectory(h, &b.io, &bad, n, (int32_t)T, 16, 2, 0, nullptr) == SCG_ERR_INVALID); }
    { scg_trajectory_sink bad = sink; bad.obs = state.data(); EXPECT(scg_rollout_trajectory(h, &b.io, &bad, n, (int32_t)T, 16, 2, 0, nullptr) == SCG_ERR_INVALID); }
    { scg_trajectory_sink bad = sink; bad.constraint_values = srew.data(); bad.constraint_values_step_stride = ld;
      EXPECT(scg_rollout_trajectory(h, &b.io, &bad, n, (int32_t)T, 16, 2, 0, nullptr) == SCG_ERR_INVALID); }          // no constraints attached
    { Buffers x(ld); x.io.action_rows = 1; EXPECT(scg_rollout_trajectory(h, &x.io, &sink, n, (int32_t)T, 16, 2, 0, nullptr) == SCG_ERR_INVALID); }
    shim_reset();
    EXPECT(scg_rollout_trajectory(h, &b.io, &sink, n, 0, 16, 2, 0, nullptr) == SCG_OK && g_shim.launches == 0);
    EXPECT(scg_rollout_trajectory(h, &b.io, &sink, n, (int32_t)T, 16, 2, 0, nullptr) == SCG_OK);
    EXPECT(g_shim.step_launches_main == T && g_shim.graph_launches == 0);
    // the last launch read slab T - 2 and wrote slab T - 1; reward / done of step T - 1
    EXPECT(g_shim.last_io.state == sobs.data() + (T - 2) * 12 * ld && g_shim.last_io.state_out == sobs.data() + (T - 1) * 12 * ld);
    EXPECT(g_shim.last_io.reward == srew.data() + (T - 1) * ld && g_shim.last_io.done == sdone.data() + (T - 1) * ld);
    EXPECT(g_shim.last_io.obs == nullptr);
    // ... and the last slab came home: (nx - 1) planes of ld + n floats
    EXPECT(g_shim.async_copies == 1 && g_shim.last_async_copy_bytes == (int64_t)((11 * ld + n) * sizeof(float)));
    sink.flags = SCG_SINK_KEEP_STATE_IN_SINK;
    shim_reset();
    EXPECT(scg_rollout_trajectory(h, &b.io, &sink, n, (int32_t)T, 16, 2, 0, nullptr) == SCG_OK && g_shim.async_copies == 0);
    sink.flags = 0;
    // a single step: read io.state, write slab 0
    shim_reset();
    EXPECT(scg_rollout_trajectory(h, &b.io, &sink, n, 1, 16, 2, 0, nullptr) == SCG_OK);
    EXPECT(g_shim.last_io.state == state.data() && g_shim.last_io.state_out == sobs.data());
    // sink without an observation member: the state stays in place
    { scg_trajectory_sink r = sink; r.obs = nullptr; shim_reset();
      EXPECT(scg_rollout_trajectory(h, &b.io, &r, n, (int32_t)T, 16, 2, 0, nullptr) == SCG_OK);
      EXPECT(g_shim.last_io.state == state.data() && g_shim.last_io.state_out == nullptr && g_shim.async_copies == 0); }
    // a signature seen twice is NOT graphed with a sink
    shim_reset();
    for (int k = 0; k < 3; ++k) EXPECT(scg_rollout_trajectory(h, &b.io, &sink, n, 16, 16, 2, 0, nullptr) == SCG_ERR_INVALID || true);
    EXPECT(g_shim.graph_launches == 0 && g_shim.captured_nodes == 0);
    // the fused form: one launch carrying the sink
    shim_reset();
    EXPECT(scg_rollout_trajectory(h, &b.io, &sink, n, (int32_t)T, 16, 2, 1, nullptr) == SCG_OK);
    EXPECT(g_shim.fused_launches == 1 && g_shim.last_fused_sink_obs == sobs.data() && g_shim.async_copies == 0 && g_shim.step_launches_main == 0);
    // ... which a per-step info sink (truncated / mse) cannot take: per-step launches
    { scg_trajectory_sink r = sink; r.mse = srew.data(); r.mse_step_stride = ld; shim_reset();
      EXPECT(scg_rollout_trajectory(h, &b.io, &r, n, (int32_t)T, 16, 2, 1, nullptr) == SCG_OK && g_shim.fused_launches == 0 && g_shim.step_launches_main == T);
      EXPECT(g_shim.last_io.mse == srew.data() + (T - 1) * ld); }
    EXPECT(scg_destroy(h) == SCG_OK);
    // two launch queues with a sink: the second half sees every pointer shifted by `half`, and is joined
    c.rollout_split_min = 256;
    EXPECT(scg_create(&c, &h) == SCG_OK);
    shim_reset();
    const int64_t half = (n / 2) / 256 * 256;
    EXPECT(scg_rollout_trajectory(h, &b.io, &sink, n, (int32_t)T, 16, 2, 0, nullptr) == SCG_OK);
    EXPECT(g_shim.step_launches_main == T && g_shim.step_launches_aux == T && g_shim.aux_streams_open == 0);
    EXPECT(g_shim.last_io.state_out == sobs.data() + (T - 1) * 12 * ld + half && g_shim.last_io.reward == srew.data() + (T - 1) * ld + half);
    EXPECT(g_shim.async_copies == 1);
    for (int k = 0; k < 6; ++k) {                                            // a failing launch still joins the second queue
        shim_reset();
        g_shim.fail_launch_after = k;
        EXPECT(scg_rollout_trajectory(h, &b.io, &sink, n, (int32_t)T, 16, 2, 0, nullptr) == SCG_ERR_HIP && g_shim.aux_streams_open == 0 && g_shim.async_copies == 0);
    }
    shim_reset();
    EXPECT(scg_destroy(h) == SCG_OK);
}

// An exec whose last replay cannot be waited for is leaked, never destroyed under a running replay.
static void check_graph_retirement() {
    scg_config c = base_config(SCG_QUAD_2D);
    scg_handle* h = nullptr;
    Buffers b(4096);
    EXPECT(scg_create(&c, &h) == SCG_OK);
    shim_reset();
    EXPECT(scg_rollout(h, &b.io, 4096, 16, 16, 2, nullptr) == SCG_OK && scg_rollout(h, &b.io, 4096, 16, 16, 2, nullptr) == SCG_OK);
    EXPECT(g_shim.execs_live == 1 && g_shim.graph_launches == 1);
    const int64_t syncs = g_shim.event_syncs;
    EXPECT(scg_set_constraints(h, nullptr, 0, nullptr, 0) == SCG_OK);        // drops the graphs: waits for the entry's last replay only
    EXPECT(g_shim.execs_live == 0 && g_shim.event_syncs == syncs + 1);
    EXPECT(scg_rollout(h, &b.io, 4096, 16, 16, 2, nullptr) == SCG_OK && scg_rollout(h, &b.io, 4096, 16, 16, 2, nullptr) == SCG_OK);
    EXPECT(g_shim.execs_live == 1);
    g_shim.fail_event_sync = 1;
    EXPECT(scg_set_constraints(h, nullptr, 0, nullptr, 0) == SCG_OK);
    EXPECT(g_shim.execs_live == 1);                                           // leaked on purpose
    g_shim.fail_event_sync = 0;
    EXPECT(hipGraphExecDestroy(static_cast<hipGraphExec_t>(g_shim.last_exec)) == hipSuccess && g_shim.execs_live == 0);   // the test frees what the layer leaked
    EXPECT(scg_destroy(h) == SCG_OK);
}

static void check_next_row_entry_points() {
    std::vector<float> f(64);
    std::vector<double> d(64);
    std::vector<uint8_t> u(64);
    scg_config c = base_config(SCG_QUAD_2D);
    scg_handle* h = nullptr;
    EXPECT(scg_create(&c, &h) == SCG_OK);
    EXPECT(scg_reset(nullptr, f.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 8, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_reset(h, f.data(), nullptr, (int32_t*)f.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 4, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_reset(h, f.data(), nullptr, (int32_t*)f.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 8, 8, nullptr) == SCG_OK);
    EXPECT(scg_eval_constraints(h, nullptr, nullptr, f.data(), nullptr, 8, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_eval_constraints(h, f.data(), nullptr, f.data(), nullptr, 4, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_model_eval(h, f.data(), f.data(), nullptr, nullptr, nullptr, 8, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_model_eval(h, f.data(), f.data(), nullptr, f.data(), nullptr, 8, 8, nullptr) == SCG_OK);
    EXPECT(scg_model_flow(h, d.data(), d.data(), nullptr, 0.02, 1e-6, 1e-8, 8, nullptr, nullptr, nullptr, nullptr, 8, 8, nullptr) == SCG_ERR_INVALID);   // no output
    EXPECT(scg_model_flow(h, d.data(), d.data(), nullptr, 0.0, 1e-6, 1e-8, 8, d.data(), nullptr, nullptr, nullptr, 8, 8, nullptr) == SCG_ERR_INVALID);       // dt
    EXPECT(scg_model_flow(h, d.data(), d.data(), nullptr, 0.02, 0.0, 0.0, 8, d.data(), nullptr, nullptr, nullptr, 8, 8, nullptr) == SCG_ERR_INVALID);        // tolerances
    EXPECT(scg_model_flow(h, d.data(), d.data(), nullptr, 0.02, 1e-6, 1e-8, 17, d.data(), nullptr, nullptr, nullptr, 8, 8, nullptr) == SCG_ERR_INVALID);     // doublings
    EXPECT(scg_model_flow(h, d.data(), d.data(), nullptr, 0.02, 1e-6, 1e-8, 8, d.data(), nullptr, nullptr, nullptr, 4, 8, nullptr) == SCG_ERR_INVALID);      // ld < n
    EXPECT(scg_model_flow(h, d.data(), d.data(), nullptr, 0.02, 1e-6, 1e-8, 8, nullptr, d.data(), nullptr, nullptr, ((int64_t)1 << 26) + 1, ((int64_t)1 << 26) + 1, nullptr) == SCG_ERR_INVALID);   // n with sensitivities
    EXPECT(scg_model_flow(h, d.data(), d.data(), nullptr, 0.02, 1e-6, 1e-8, 8, d.data(), d.data(), d.data(), (int32_t*)f.data(), 8, 8, nullptr) == SCG_OK);
    EXPECT(scg_model_flow(h, d.data(), d.data(), nullptr, 0.02, 1e-6, 0.0, 0, d.data(), nullptr, nullptr, nullptr, 8, 0, nullptr) == SCG_OK);                 // n == 0
    EXPECT(scg_linearize(h, f.data(), f.data(), nullptr, f.data(), f.data(), -1.0, 8, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_linearize(h, f.data(), f.data(), nullptr, f.data(), f.data(), 0.02, 8, 8, nullptr) == SCG_OK);
    EXPECT(scg_fc_linear(h, nullptr, nullptr, f.data(), f.data(), nullptr, f.data(), 8, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_fc_linear(h, nullptr, f.data(), f.data(), f.data(), nullptr, f.data(), 8, 8, nullptr) == SCG_OK);
    EXPECT(scg_discretize_exact(f.data(), f.data(), 12, 5, 0.02, f.data(), f.data(), nullptr, nullptr, nullptr, 4, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_discretize_exact(f.data(), f.data(), 6, 2, 0.0, f.data(), f.data(), nullptr, nullptr, nullptr, 4, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_discretize_exact(f.data(), f.data(), 6, 2, 0.02, nullptr, nullptr, nullptr, nullptr, nullptr, 4, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_discretize_exact(f.data(), f.data(), 6, 2, 0.02, nullptr, nullptr, nullptr, nullptr, f.data(), 4, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_discretize_exact(f.data(), f.data(), 6, 2, 0.02, f.data(), f.data(), f.data(), nullptr, f.data(), 4, nullptr) == SCG_OK);
    EXPECT(scg_quadratic_loss(h, d.data(), d.data(), f.data(), f.data(), f.data(), 4, f.data(), 0, f.data(), nullptr, nullptr, 8, 8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_quadratic_loss(h, d.data(), d.data(), f.data(), f.data(), f.data(), 0, f.data(), 0, f.data(), nullptr, nullptr, 8, 8, nullptr) == SCG_OK);
    EXPECT(scg_gae(f.data(), f.data(), f.data(), nullptr, nullptr, f.data(), f.data(), f.data(), nullptr, nullptr, 0, 8, 8, 0.99f, 0.95f, 1, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_gae(f.data(), f.data(), f.data(), nullptr, nullptr, f.data(), f.data(), f.data(), f.data(), nullptr, 4, 8, 8, 0.99f, 0.95f, 1, nullptr) == SCG_OK);
    // exactly one of the float mask planes and the step kernel's done byte planes
    EXPECT(scg_gae(f.data(), f.data(), nullptr, u.data(), nullptr, f.data(), f.data(), f.data(), f.data(), nullptr, 4, 8, 8, 0.99f, 0.95f, 1, nullptr) == SCG_OK);
    EXPECT(scg_gae(f.data(), f.data(), f.data(), u.data(), nullptr, f.data(), f.data(), f.data(), f.data(), nullptr, 4, 8, 8, 0.99f, 0.95f, 1, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_gae(f.data(), f.data(), nullptr, nullptr, nullptr, f.data(), f.data(), f.data(), f.data(), nullptr, 4, 8, 8, 0.99f, 0.95f, 1, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_normalize_advantages(f.data(), nullptr, 4, 8, 8, 1e-6f, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_mean_std_normalize(f.data(), f.data(), 0, 8, 8, 8, d.data(), d.data() + 32, d.data(), 10.f, 1e-8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_mean_std_normalize(f.data(), f.data(), 3, 8, 8, 8, d.data(), d.data(), d.data(), 10.f, 1e-8, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_mean_std_normalize(f.data(), f.data(), 3, 8, 8, 8, d.data(), d.data() + 32, d.data(), 10.f, 1e-8, nullptr) == SCG_OK);
    EXPECT(scg_reward_normalize(f.data(), nullptr, nullptr, f.data(), 8, d.data(), d.data() + 8, d.data(), 0.99f, 10.f, 1e-8, 0, nullptr) == SCG_ERR_INVALID);
    EXPECT(scg_reward_normalize(f.data(), u.data(), f.data(), f.data(), 8, d.data(), d.data() + 8, d.data(), 0.99f, 10.f, 1e-8, 0, nullptr) == SCG_OK);
    // the two passes apart (statistics shared by several shards): argument checks of the four entry points
    {
        double* rec0 = d.data(); double* rec1 = d.data() + 16; double* scratch = d.data() + 32; double* sums = d.data() + 48;
        EXPECT(scg_mean_std_moments(f.data(), 2, 4, 4, rec0, scratch, nullptr, nullptr) == SCG_ERR_INVALID);           // no sums
        EXPECT(scg_mean_std_moments(f.data(), 2, 4, 0, rec0, scratch, sums, nullptr) == SCG_ERR_INVALID);              // empty shard
        EXPECT(scg_mean_std_moments(f.data(), 2, 3, 4, rec0, scratch, sums, nullptr) == SCG_ERR_INVALID);              // ld < n
        EXPECT(scg_mean_std_moments(f.data(), 2, 4, 4, rec0, scratch, sums, nullptr) == SCG_OK);
        EXPECT(scg_mean_std_normalize_sums(f.data(), f.data(), 2, 4, 4, 4, rec0, rec0, sums, 8.0, 10.f, 1e-8, nullptr) == SCG_ERR_INVALID);   // aliasing records
        EXPECT(scg_mean_std_normalize_sums(f.data(), f.data(), 2, 4, 4, 4, rec0, rec1, sums, 3.0, 10.f, 1e-8, nullptr) == SCG_ERR_INVALID);   // fewer rows than this shard's
        EXPECT(scg_mean_std_normalize_sums(f.data(), f.data(), 2, 4, 4, 4, rec0, rec1, nullptr, 8.0, 10.f, 1e-8, nullptr) == SCG_ERR_INVALID);
        EXPECT(scg_mean_std_normalize_sums(f.data(), f.data(), 2, 4, 4, 4, rec0, rec1, sums, 8.0, 10.f, 1e-8, nullptr) == SCG_OK);
        EXPECT(scg_reward_moments(f.data(), nullptr, f.data(), 8, rec0, scratch, sums, 0.99f, nullptr) == SCG_ERR_INVALID);
        EXPECT(scg_reward_moments(f.data(), u.data(), f.data(), 8, rec0, scratch, sums, 0.99f, nullptr) == SCG_OK);
        EXPECT(scg_reward_normalize_sums(f.data(), f.data(), 8, rec0, rec0, sums, 16.0, 10.f, 1e-8, nullptr) == SCG_ERR_INVALID);
        EXPECT(scg_reward_normalize_sums(f.data(), f.data(), 8, rec0, rec1, sums, 7.0, 10.f, 1e-8, nullptr) == SCG_ERR_INVALID);
        EXPECT(scg_reward_normalize_sums(f.data(), f.data(), 8, rec0, rec1, sums, 16.0, 10.f, 1e-8, nullptr) == SCG_OK);
    }
    {
        // scg_copy_segments (PPOBuffer.push in one launch): count bounds, null pointers, negative / aliasing strides, empty segments
        scg_copy_segment seg[SCG_MAX_COPY_SEGMENTS + 1] = {};
        for (auto& g : seg) g = scg_copy_segment{f.data(), f.data() + 32, 4, 2, 2, 1, 2, 1};
        EXPECT(scg_copy_segments(nullptr, 1, nullptr) == SCG_ERR_INVALID);
        EXPECT(scg_copy_segments(seg, 0, nullptr) == SCG_ERR_INVALID);
        EXPECT(scg_copy_segments(seg, SCG_MAX_COPY_SEGMENTS + 1, nullptr) == SCG_ERR_INVALID);
        EXPECT(scg_copy_segments(seg, SCG_MAX_COPY_SEGMENTS, nullptr) == SCG_OK);
        seg[1].dst = nullptr;
        EXPECT(scg_copy_segments(seg, 2, nullptr) == SCG_ERR_INVALID);
        seg[1].rows = 0;                                                      // ... an empty segment may carry null pointers
        EXPECT(scg_copy_segments(seg, 2, nullptr) == SCG_OK);
        seg[0].src_row_stride = -2;
        EXPECT(scg_copy_segments(seg, 1, nullptr) == SCG_ERR_INVALID);
        seg[0].src_row_stride = 0;                                            // a broadcast source is fine
        EXPECT(scg_copy_segments(seg, 1, nullptr) == SCG_OK);
        seg[0].dst_row_stride = 0;                                            // an aliasing destination is not
        EXPECT(scg_copy_segments(seg, 1, nullptr) == SCG_ERR_INVALID);
        seg[0].dst_row_stride = 2; seg[0].rows = -1;
        EXPECT(scg_copy_segments(seg, 1, nullptr) == SCG_ERR_INVALID);
        seg[0].rows = (int64_t)1 << 41;
        EXPECT(scg_copy_segments(seg, 1, nullptr) == SCG_ERR_INVALID);
    }
    EXPECT(scg_destroy(h) == SCG_OK);
}

int main() {
    shim_reset();
    check_create_validation();
    for (int system = SCG_QUAD_1D; system <= SCG_CARTPOLE; ++system) check_tables(system);
    check_step_and_rollout();
    check_rollout_graphs();
    check_graph_retirement();
    check_async_step();
    check_device_guard();
    check_trajectory_sink();
    check_next_row_entry_points();
    EXPECT(g_shim.live_allocations == 0);
    std::printf(g_fail ? "capi sanitize driver: %d expectation(s) FAILED\n" : "capi sanitize driver: ok\n", g_fail);
    return g_fail ? 1 : 0;
}
