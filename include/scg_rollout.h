/*
 * scg_rollout.h -- C ABI of the MI355X-native vectorised rollout engine that replaces the
 * env.step() hot path of ekumenlabs/safe-control-gym.
 *
 * The reference is 100 % Python: it has no FFI for this path.  The entry points
 * below are what a binding for the path would bind; each cites the reference interface it
 * replaces (file:line relative to /root/reference/safe_control_gym).  INTEGRATION.md shows the
 * ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C, no C++/torch types; every array argument of scg_step/scg_reset/scg_rollout is a
 *     DEVICE pointer owned by the caller (e.g. torch.Tensor.data_ptr() of a ROCm tensor);
 *     the handle owns only its constant tables (reference trajectory, constraint rows, disturbance entries).
 *   - struct-of-arrays layout: a "plane" is N consecutive values of one state/action/constraint
 *     component; plane j of an array starts at base + j*ld elements (ld >= n, the plane stride).
 *     Row-major exceptions, each said where it is declared: scg_step_io.terminal_record, the action when
 *     scg_step_io.action_rows = 1, and the Jacobian batches of scg_linearize.
 *   - every launch is asynchronous on the hipStream_t passed in; no implicit synchronisation.
 *   - return value: 0 on success, negative scg_status on error; nothing throws across the ABI;
 *     scg_last_error() returns a thread-local message for the last failing call.
 *   - a handle is not thread-safe; distinct handles are.  One handle per device: a handle belongs to the device that was
 *     current in scg_create(), and every call that takes a handle must be made with THAT device current -- anything else
 *     returns SCG_ERR_INVALID naming both device indices (the handle's tables live on its device; a launch from another
 *     one would read them across the fabric, or fault).  The handle-free entry points (scg_gae, scg_normalize_advantages,
 *     scg_discretize_exact, the normalisers) run on whatever device is current, like any HIP call.
 */
#ifndef SCG_ROLLOUT_H
#define SCG_ROLLOUT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCG_ABI_VERSION 6
#define SCG_MAX_NX 12        /* quadrotor.py:625  (3-D: nx = 12) */
#define SCG_MAX_NU 4         /* quadrotor.py:625  (3-D: nu = 4)  */
#define SCG_MAX_NV 16        /* constrained variable: [state ; raw action], constraints.py:58-63 */
#define SCG_MAX_CONSTRAINT_ROWS 64
#define SCG_MAX_QUADRATIC 4
#define SCG_MAX_GOAL_HORIZON 8
#define SCG_MAX_DISTURBANCES 4       /* per disturbance mode */
#define SCG_STATS_STRIPES 64         /* rows of scg_step_io.stats */

typedef enum scg_status {
    SCG_OK = 0,
    SCG_ERR_INVALID = -1,       /* bad argument / config (the Python side raises ValueError first) */
    SCG_ERR_HIP = -2,           /* a HIP runtime call failed; see scg_last_error() */
    SCG_ERR_NOMEM = -3,
    SCG_ERR_UNSUPPORTED = -4
} scg_status;

/* QuadType, envs/gym_pybullet_drones/quadrotor_utils.py:11-18; CARTPOLE per benchmark_env.py:43 */
typedef enum scg_system { SCG_QUAD_1D = 1, SCG_QUAD_2D = 2, SCG_QUAD_3D = 3, SCG_CARTPOLE = 4 } scg_system;
/* controllers/mpc/mpc_utils.py:79-101 (RK4); explicit Euler as the cheap alternative */
typedef enum scg_integrator { SCG_RK4 = 0, SCG_EULER = 1 } scg_integrator;
/* envs/benchmark_env.py:29-35 */
typedef enum scg_task { SCG_STABILIZATION = 0, SCG_TRAJ_TRACKING = 1 } scg_task;
/* envs/benchmark_env.py:19-26 (COMPETITION needs contact queries: out of scope) */
typedef enum scg_cost { SCG_RL_REWARD = 0, SCG_QUADRATIC = 1 } scg_cost;

typedef struct scg_handle scg_handle;

/* Everything Quadrotor.__init__ / BenchmarkEnv.__init__ fix for the lifetime of an env
 * (quadrotor.py:148-329, benchmark_env.py:63-218, base_aviary.py:66-147).  Doubles on purpose:
 * derived constants (1/m, l/sqrt2/Iyy, KM/KF ...) are formed in double and rounded to float once. */
typedef struct scg_config {
    uint32_t struct_size;            /* = sizeof(scg_config) */
    uint32_t abi_version;            /* = SCG_ABI_VERSION */
    int32_t system;                  /* scg_system */
    int32_t integrator;              /* scg_integrator */
    int32_t substeps;                /* fixed steps of dt/substeps per step(); 1 = mpc.py:163 usage,
                                        PYB_STEPS_PER_CTRL mirrors base_aviary.py:278 */
    int32_t task;                    /* scg_task */
    int32_t cost;                    /* scg_cost */
    int32_t ctrl_steps;              /* CTRL_STEPS, benchmark_env.py:161 (time limit :460) */
    double dt;                       /* CTRL_TIMESTEP, benchmark_env.py:157 */
    /* quadrotor physical constants: assets/cf2x.urdf:5,12,13; base_aviary.py:90 */
    double mass, gravity, arm, ixx, iyy, izz, kf, km;
    double pwm2rpm_scale, pwm2rpm_const, pwm_min, pwm_max;
    /* cartpole (not in the reference fork; names from tests/scripts/verbose_api.yaml:32-39) */
    double pole_length, pole_mass, cart_mass;
    /* action preprocessing: quadrotor.py:815-842, quadrotor_utils.py:21-67 */
    int32_t normalized_action;       /* NORMALIZED_RL_ACTION_SPACE */
    int32_t motor_map;               /* apply cmd2pwm -> pwm2rpm -> KF rpm^2 */
    double norm_act_scale, hover_thrust, action_scale;
    double act_low[SCG_MAX_NU], act_high[SCG_MAX_NU];   /* action_space bounds (float32-rounded) */
    /* goals and cost: quadrotor.py:262-329,886-930, benchmark_env.py:190-192 */
    double u_goal[SCG_MAX_NU];
    double x_goal[SCG_MAX_NX];       /* stabilisation goal; tracking uses scg_set_reference() */
    double rew_state_weight[SCG_MAX_NX], rew_act_weight[SCG_MAX_NU];
    int32_t rew_exponential;
    int32_t reserved0;
    double Q[SCG_MAX_NX * SCG_MAX_NX];   /* row-major nx*nx in the top-left corner of a 12-wide matrix */
    double R[SCG_MAX_NU * SCG_MAX_NU];   /* row-major, 4-wide */
    double mse_weight[SCG_MAX_NX];       /* info_mse_metric_state_weight, quadrotor.py:185-200 */
    /* termination: quadrotor.py:956-993, benchmark_env.py:435-462 */
    int32_t done_on_out_of_bound, done_on_violation;
    double oob_low[SCG_MAX_NX], oob_high[SCG_MAX_NX];   /* +-inf on dims the mask (:981-985) excludes */
    double goal_tolerance;           /* stabilization_goal_tolerance, quadrotor.py:138,965 */
    int32_t use_constraint_penalty;  /* benchmark_env.py:454-456 */
    int32_t obs_goal_horizon;        /* benchmark_env.py:406-420 */
    double constraint_penalty;
    /* reset: quadrotor.py:478-493, benchmark_env.py:277-308 (distribution per dim: init_rand_kind below) */
    int32_t randomized_init;
    int32_t auto_reset;              /* dummy_vec_env.py:41-47 fused into step */
    double init_state[SCG_MAX_NX];
    double init_rand_low[SCG_MAX_NX], init_rand_high[SCG_MAX_NX];  /* low == high == 0: dim not randomised */
    /* per-env inertial randomisation: quadrotor.py:455-467 (value = nominal + U(low, high) per reset) over
     * (M, Ixx, Iyy, Izz); cartpole: (pole_length, pole_mass, cart_mass, -) per tests/scripts/verbose_api.yaml:40-54 */
    int32_t randomized_inertial;
    int32_t reserved2;
    double inertial_rand_low[4], inertial_rand_high[4];
    /* Physics.PYB_GND / PYB_DRAG (base_aviary.py:35-45,437-500) as zero-order-hold terms of the analytic model
     * (quadrotor only): ground effect -- every propeller's thrust scaled by 1 + gnd_eff_coeff (prop_radius / (4 h_i))^2,
     * h_i = max(height of propeller i, gnd_eff_h_clip), while |roll|, |pitch| < pi/2 (:459-463), re-evaluated at every
     * integration sub-step; drag -- the reference's R (R (k * v)) force with k = -drag_coeff * sum_i 2 pi rpm_i / 60
     * (:487-490; applied in the link frame, hence the second rotation), held over the control step.  Downwash
     * (:498-520) acts between drones of one env and does not exist for a single drone. */
    int32_t ground_effect, drag;
    double gnd_eff_coeff, prop_radius, gnd_eff_h_clip;
    double drag_coeff_xy, drag_coeff_z;
    uint64_t seed;                   /* Philox key */
    int64_t env_index_base;          /* global index of this shard's env 0 (seed + rank analogue,
                                        env_wrappers/vectorized_env/__init__.py:29) */
    /* Distribution of every randomised quantity: benchmark_env.py:297-307 calls np_random.<distrib>(*args, **kwargs)
     * for whatever name the yaml gives.  kind = SCG_RAND_UNIFORM keeps the low / high fields above; the other kinds take
     * their parameters from *_rand_p[dim][0..2] (scg_distribution below).  Value = nominal + draw, as the reference. */
    int32_t init_rand_kind[SCG_MAX_NX];
    double init_rand_p[SCG_MAX_NX][3];
    int32_t inertial_rand_kind[4];
    double inertial_rand_p[4][3];
    /* Launch policy of THIS handle (no process-wide switches: two handles in one process may differ).
     * rollout_queues: 0 = default (2), 1 = everything on the caller's stream, 2 = scg_rollout cuts batches whose half is
     *   >= rollout_split_min envs in two halves advancing on two HIP streams.
     * rollout_split_min: 0 = default (262144 envs per half: batches of at least 2^19 envs; below, the graph replay is level or ahead).
     * ref_in_lds: stage the X_GOAL table in LDS once per workgroup instead of gathering it through L1/L2.
     * grid_cap_per_cu: 0 = one lane per env (no grid-stride loop); > 0 caps the grid at that many workgroups per CU.
     * rollout_graph: 0 = default (on), 1 = on, -1 = off.  On: scg_rollout captures the launches of a single-queue rollout
     *   (batches below the two-queue threshold, >= 8 steps) into a hipGraph the second time it sees the same signature
     *   (same scg_step_io pointers, n, steps, action layout) and replays it from then on; see scg_rollout. */
    int32_t rollout_queues;
    int32_t ref_in_lds;
    int64_t rollout_split_min;
    int32_t grid_cap_per_cu;
    int32_t rollout_graph;
    /* lone_launch_forms: 0 = default (on), 1 = on, -1 = off.  On: a step issued by scg_step / scg_step_async -- ONE dependent launch,
     *   a policy between it and the next -- runs the kernel forms that measured faster for a launch with nothing of its kind
     *   overlapping it (per-wave instead of workgroup-pooled reset draws: no workgroup barrier; plain instead of nontemporal
     *   terminal-record stores), the launches queued by scg_rollout keep the forms that are faster back to back; 1 extends the
     *   lone forms to scg_rollout's single-queue launches (dependent launches on one queue do not overlap either).  Results are
     *   bit-identical either way. */
    int32_t lone_launch_forms;
    int32_t reserved3;
} scg_config;

/* np.random.RandomState methods the reset path can draw from on the device (benchmark_env.py:297-307); u, u2 are
 * Philox uniforms in [0, 1).  Multivariate draws, vonmises / f / standard_t ... and choice() over more than three values are rejected by the
 * Python layer. */
typedef enum scg_distribution {
    SCG_RAND_UNIFORM = 0,        /* uniform(low, high): low + (high - low) u              (fields *_rand_low / high) */
    SCG_RAND_NORMAL = 1,         /* normal(loc = p0, scale = p1): Box-Muller on (u, u2); standard_normal: p = (0, 1) */
    SCG_RAND_EXPONENTIAL = 2,    /* exponential(scale = p0): -p0 ln(1 - u) */
    SCG_RAND_LAPLACE = 3,        /* laplace(loc = p0, scale = p1): inverse CDF */
    SCG_RAND_LOGISTIC = 4,       /* logistic(loc = p0, scale = p1): p0 + p1 ln(u / (1 - u)) */
    SCG_RAND_TRIANGULAR = 5,     /* triangular(left = p0, mode = p1, right = p2): inverse CDF */
    SCG_RAND_RANDINT = 6,        /* randint(low = p0, high = p1): floor(p0 + (p1 - p0) u) */
    SCG_RAND_CHOICE = 7,         /* choice(a) over 1..3 equiprobable values p0, p1, p2: p[floor(count u)], count in *_rand_low */
    SCG_RAND_RAYLEIGH = 8,       /* rayleigh(scale = p0): p0 sqrt(-2 ln(1 - u)) */
    SCG_RAND_GUMBEL = 9,         /* gumbel(loc = p0, scale = p1): p0 - p1 ln(-ln u) */
    SCG_RAND_WEIBULL = 10,       /* weibull(a = p0): (-ln(1 - u))^(1 / p0) */
    SCG_RAND_PARETO = 11,        /* pareto(a = p0) (Lomax, as numpy): (1 - u)^(-1 / p0) - 1 */
    SCG_RAND_POWER = 12,         /* power(a = p0): u^(1 / p0) */
    SCG_RAND_LOGNORMAL = 13,     /* lognormal(mean = p0, sigma = p1): exp(normal(p0, p1)) on (u, u2) */
    SCG_RAND_CAUCHY = 14,        /* standard_cauchy(): tan(pi (u - 1/2)) */
    /* Rejection samplers: a bounded loop over Philox blocks of their own, counter (env, serial, word0 + attempt) with word0 = 256 + 16 dim
     * (initial state) / 512 + 16 q (inertial property) -- the draw of one quantity never shifts another's.  8 attempts per gamma variate
     * (Marsaglia-Tsang accepts with probability > 0.95: all 8 fail once in 1e10 draws, the last candidate is then used). */
    SCG_RAND_GAMMA = 15,         /* gamma(shape = p0, scale = p1) (standard_gamma: p1 = 1; chisquare(df): shape df / 2, scale 2): Marsaglia-Tsang
                                    on (normal, uniform) pairs; shape < 1 by the boost gamma(shape + 1) u^(1 / shape) */
    SCG_RAND_BETA = 16,          /* beta(a = p0, b = p1): x / (x + y) of two gamma variates (words word0 .. +7 and word0 + 8 .. +15) */
    SCG_RAND_CHOICE_P = 17       /* choice(a, p = ...) over up to three values p0, p1, p2 with probabilities: *_rand_low = P(p0),
                                    *_rand_high = P(p0) + P(p1) (cumulative; two values: high = 1): u < low ? p0 : u < high ? p1 : p2 */
} scg_distribution;
#define SCG_RAND_KIND_MAX 17

/* One scalar constraint g(v) <= 0 over v = [state ; RAW action] (constraints.py:160-172).
 * LinearConstraint / BoundedConstraint / DefaultConstraint (constraints.py:226-368) compile to
 * kind 0 rows with coef = (A @ filter) scattered to v's index space;
 * QuadraticContstraint (constraints.py:179-223) to one kind 1 row with P = filter' P filter. */
typedef struct scg_constraint_row {
    int32_t kind;                    /* 0: c = coef . v - b     1: c = v' P[p_index] v - b */
    int32_t strict;                  /* constraints.py:137-141 */
    int32_t has_tolerance;           /* constraints.py:153-158 */
    int32_t p_index;
    double tolerance;
    double b;
    double coef[SCG_MAX_NV];
} scg_constraint_row;

/* Device buffers of one step() call.  Nullable members are skipped (their bytes are not moved). */
typedef struct scg_step_io {
    uint32_t struct_size;            /* = sizeof(scg_step_io) */
    uint32_t action_rows;            /* 0: action is SoA planes [nu][ld]; 1: row-major [n][nu] (what a policy emits), read as
                                        one 4 nu-byte vector per lane; must be 4 nu-byte aligned; scg_step only */
    int64_t ld;                      /* plane stride in elements (>= n) of every multi-plane array */
    /* in */
    const float* action;             /* [nu][ld] (or [n][nu], see action_rows): raw action as passed to step() */
    const float* action_disturbance; /* nullable [nu][ld], added after clipping (quadrotor.py:835-838) */
    const float* dynamics_force;     /* nullable [3][ld] world-frame force (quadrotor.py:526-548) */
    const float* obs_noise;          /* nullable [nx][ld] added to obs only (quadrotor.py:879-880) */
    const float* reset_state;        /* nullable [nx][ld] per-env nominal reset state (overrides cfg.init_state) */
    /* in/out */
    float* inertial;                 /* nullable [4][ld] per-env (M, Ixx, Iyy, Izz) | (pole_length, pole_mass,
                                        cart_mass, -); read every step, re-drawn on auto-reset when
                                        cfg.randomized_inertial */
    float* state;                    /* [nx][ld] */
    float* state_lo;                 /* nullable [nx][ld] (or [6][ld], see state_lo_kind) compensation residual (state = hi + lo) */
    float* state_out;                /* nullable [nx][ld]: where the new state is written; NULL (or == state) = in place.  With
                                        a buffer of its own `state` is only read: a caller that hands every step's observation
                                        out as a tensor of its own (the reference returns fresh arrays) passes a new buffer
                                        here, keeps it as the next step's `state`, and moves no byte twice.  scg_step only */
    int32_t* step_count;             /* [n] ctrl_step_counter */
    float* ep_return;                /* nullable [n] running episode return (record_episode_statistics.py:145) */
    int32_t* ep_length;              /* nullable [n]; NULL: the episode length is read off step_count (the two are equal
                                        whenever finished envs are reset, i.e. with cfg.auto_reset) */
    uint32_t* episode_id;            /* nullable [n] serial of the reset that began the env's episode; needed only
                                        when a disturbance draws its step_offset per episode */
    /* out */
    float* obs;                      /* nullable [obs_dim][ld]; NULL when the observation IS the state (no goal horizon, no
                                        observation noise).  obs == state (== state_out if given) is allowed with a goal horizon and no observation
                                        noise: the buffer then holds obs_dim planes, the first nx of which are the state, and
                                        the kernel writes the state once and only the goal rows behind it */
    float* reward;                   /* [n] */
    uint8_t* done;                   /* [n] */
    float* constraint_values;        /* nullable [nc][ld] */
    uint8_t* constraint_violation;   /* nullable [n] */
    uint8_t* truncated;              /* nullable [n] info["TimeLimit.truncated"] */
    uint8_t* goal_reached;           /* nullable [n] */
    float* mse;                      /* nullable [n] */
    float* terminal_record;          /* nullable [n][terminal_stride] row-major, written for done lanes only -- one dense
                                        record per env instead of obs_dim + 2 scattered 4-byte stores (each of those is a
                                        read-modify-write of a whole DRAM burst):
                                        [0, obs_dim)  the observation step() would have returned without the auto-reset
                                                      (info['terminal_observation'], dummy_vec_env.py:43);
                                        [obs_dim]     the finished episode's return (float), [obs_dim + 1] its length
                                                      (int32 bits) -- info['episode']['r' / 'l']
                                                      (record_episode_statistics.py:155-158); 0 without ep_return */
    int64_t terminal_stride;         /* floats per record: a multiple of 4, >= obs_dim + 2.  0: terminal_record is SoA planes
                                        [obs_dim + 2][ld] instead (plane j = member j of the record; done lanes only) -- for a
                                        consumer that reads the terminal observations of ALL envs every step (the critic of the
                                        time-limit bootstrap, controllers/ppo/ppo.py:264-273): 4 obs_dim bytes per env to read
                                        instead of 4 terminal_stride, paid for by obs_dim + 2 partial-line stores per finished env */
    double* stats;                   /* nullable [SCG_STATS_STRIPES][8] batch sums, accumulated with one f64 atomic per
                                        wave and statistic; the waves spread over the stripes (same-address f64 atomics
                                        serialise at ~40 ns each), the caller adds the stripes up.  Columns:
                                        0 sum reward, 1 #done, 2 #violation, 3 sum finished returns,
                                        4 sum finished lengths, 5 #truncated, 6-7 reserved */
    uint32_t reset_serial;           /* Philox counter word for resets happening in this call */
    uint32_t state_lo_kind;          /* which states state_lo holds residuals for: 0 = all nx (float planes [nx][ld]); 1 = the
                                        rotational chain of the 3-D quadrotor only, states 6..11 as float planes [6][ld] -- positions
                                        and velocities are then kept in float32, attitude and rates accumulate in float64
                                        ("attitude" precise mode: 48 B per env-step less; lean and info tiers only); 2 = the same
                                        with the residuals as int16 planes ((int16_t*)state_lo, [6][ld], 2-byte aligned): state =
                                        hi + lo 2^(e - 40) where hi = f 2^e, f in [1/2, 1), i.e. lo counts 2^-16 ulp(hi) -- 40
                                        significant bits, 24 B per env-step less again */
    const uint32_t* serial_word;     /* nullable DEVICE word added to reset_serial by the kernel.  For callers that record their own
                                        step loop in a HIP graph (policy kernels + scg_step launches, torch.cuda.graph): the launch
                                        parameters are baked into the graph's nodes, so a replay would repeat the serials -- and with
                                        them the reset draws and disturbance noise -- of the recording.  With serial_word the nodes
                                        carry only the OFFSET of each step (reset_serial = 0, 1, ... K - 1) and the caller writes the
                                        loop's current base serial into the word before every replay: results equal eager stepping
                                        bit for bit.  (scg_rollout's own small-batch graph replay does this internally and is not
                                        taken when the caller passes a word of its own.) */
} scg_step_io;

/* One entry of a DisturbanceList (envs/disturbances.py:46-84), evaluated inside the step kernel.
 * mode: where it is added -- SCG_DIST_ACTION: on the clipped thrusts before the motor map (quadrotor.py:835-836,
 * dim nu); SCG_DIST_DYNAMICS: external force on the CoM (quadrotor.py:526-548; dim 1: z (quadrotor) / cart force
 * (cartpole), 2: (x, z), 3: (x, y, z)); SCG_DIST_OBSERVATION: on the observed state (quadrotor.py:879-880, dim nx).
 * kind: ImpulseDisturbance (:89-143), StepDisturbance (:146-184), UniformNoise (:186-212), WhiteNoise (:215-248),
 * PeriodicNoise (:260-288).  Random draws are Philox4x32-10 streams keyed like the reset draws (seed; env index,
 * step serial), one stream per (mode, entry): the reference draws from env.np_random (MT19937) instead. */
enum { SCG_DIST_ACTION = 0, SCG_DIST_DYNAMICS = 1, SCG_DIST_OBSERVATION = 2 };
enum { SCG_DIST_IMPULSE = 0, SCG_DIST_STEP = 1, SCG_DIST_UNIFORM = 2, SCG_DIST_WHITE_NOISE = 3, SCG_DIST_PERIODIC = 4 };
typedef struct scg_disturbance {
    int32_t mode, kind;
    int32_t step_offset;             /* impulse / step: >= 0 fixed; -1: drawn per episode in [0, max_step) (:128-131) */
    int32_t max_step;                /* int(EPISODE_LEN_SEC / CTRL_TIMESTEP) (:116) */
    int32_t duration;                /* impulse (:119-122) */
    int32_t has_mask;
    double magnitude, decay_rate;    /* impulse / step */
    double scale, frequency;         /* periodic: scale sin(2 pi frequency t + phase), t = ctrl step * CTRL_TIMESTEP */
    double low[SCG_MAX_NX];          /* uniform: lower bounds; white_noise: std */
    double high[SCG_MAX_NX];         /* uniform: upper bounds */
    double mask[SCG_MAX_NX];         /* multiplies the noise when has_mask (:139-140) */
} scg_disturbance;

/* Replaces Quadrotor.__init__ (quadrotor.py:148) for a batch: copies constants to the device. */
int scg_create(const scg_config* cfg, scg_handle** out);
int scg_destroy(scg_handle* h);

/* X_GOAL table for Task.TRAJ_TRACKING (quadrotor.py:280-325): host pointer, row-major [rows][nx]. */
int scg_set_reference(scg_handle* h, const double* x_goal, int32_t rows);

/* Compiled ConstraintList (constraints.py:426-583, factory :594-612): host pointers.
 * p_mats is [n_p][SCG_MAX_NV][SCG_MAX_NV] row-major. */
int scg_set_constraints(scg_handle* h, const scg_constraint_row* rows, int32_t n_rows,
                        const double* p_mats, int32_t n_p);

/* Replaces Quadrotor.reset (quadrotor.py:356-504) + RecordEpisodeStatistics.reset
 * (record_episode_statistics.py:132-140) for the lanes where mask != 0 (all lanes if mask NULL).
 * Writes state (nominal + Philox uniform offsets) and, if inertial != NULL, the per-env inertial
 * properties (nominal, + Philox offsets when cfg.randomized_inertial); zeroes step_count / ep_return / ep_length, and
 * if obs != NULL the extended observation (+ the observation disturbances, quadrotor.py:879-880 via :497), if
 * episode_id != NULL the episode id (= reset_serial); if c_val != NULL the state-only constraint values are
 * NOT produced here (reset-time only_state values come from scg_eval_constraints). */
int scg_reset(scg_handle* h, float* state, float* state_lo, int32_t* step_count, float* ep_return,
              int32_t* ep_length, float* obs, const uint8_t* mask, const float* reset_state, float* inertial,
              uint32_t* episode_id, uint32_t reset_serial, int64_t ld, int64_t n, void* stream);

/* Passive disturbances (benchmark_env.py _setup_disturbances; yaml key `disturbances`): host pointer, at most
 * SCG_MAX_DISTURBANCES entries per mode, applied in list order.  n = 0 clears them. */
int scg_set_disturbances(scg_handle* h, const scg_disturbance* entries, int32_t n);

/* Replaces Quadrotor.step (quadrotor.py:507-558) + BenchmarkEnv.after_step
 * (benchmark_env.py:422-463) + the DummyVecEnv auto-reset (dummy_vec_env.py:34-49) +
 * VecRecordEpisodeStatistics.step_wait (record_episode_statistics.py:142-169) for n envs. */
int scg_step(scg_handle* h, const scg_step_io* io, int64_t n, void* stream);

/* `steps` consecutive scg_step() launches from one call: step k reads its raw actions at
 * io->action + (k % action_period) * action_stride elements and uses reset_serial + k.  This is the
 * open-loop rollout loop of the reference's harnesses (tests/scripts/verbose_api.py:71-77: sample an
 * action, step, repeat) with the action sequence pre-generated on the device; every other buffer of
 * `io` is reused by all steps, so the outputs hold the LAST step's values and ep_return / ep_length /
 * stats accumulate.  Same kernel and per-step HBM traffic as scg_step().  Batches of >= 2^18 envs are cut in two halves whose
 * step chains are issued on `stream` and on a second, handle-owned stream (forked from and joined back into `stream` by
 * events inside this call -- on error paths too), so that one half's kernel ramp-up / drain overlaps the other half's body;
 * per env the results are bit-identical to single-stream stepping.  scg_config.rollout_queues = 1 keeps everything on
 * `stream`; scg_config.rollout_split_min moves the threshold.
 * Smaller batches (launch-bound: a step kernel lasts a few microseconds): the second call with the same signature -- the same
 * scg_step_io contents apart from reset_serial, n, steps, action_stride, action_period -- captures its launches into a
 * hipGraph (captured on the handle's own queue: nothing runs during a capture), and that call and every later one with the
 * signature replay it on `stream` (the step serial travels through a device word, so the Philox counters are those of plain
 * launches and the results are bit-identical).  At most 4 signatures are kept (least recently used goes); scg_set_reference /
 * _constraints / _disturbances drop them.  Not taken when `stream` is itself being captured (the launches then join the
 * caller's graph), when steps < 8, or with scg_config.rollout_graph = -1; a capture that fails is ended, discarded, and the
 * handle stays on plain launches. */
int scg_rollout(scg_handle* h, const scg_step_io* io, int64_t n, int32_t steps, int64_t action_stride,
                int32_t action_period, void* stream);

/* Same contract and same results as scg_rollout(), but ONE launch: each lane keeps its env in registers for all
 * `steps` steps and reads only the step's action plane (4*nu bytes per env-step instead of 8*nx + 4*nu + 13).  Taken
 * for the lean configurations (no disturbance planes / separate obs / per-step info outputs / precise mode); anything
 * else silently runs scg_rollout().  The per-step Gym contract (obs, reward, done of EVERY step) is not produced:
 * this is for open-loop rollouts (random-action benchmarks, shooting), not for a policy in the loop. */
int scg_rollout_fused(scg_handle* h, const scg_step_io* io, int64_t n, int32_t steps, int64_t action_stride,
                      int32_t action_period, void* stream);

/* ---- the step as the reference's VecEnv protocol splits it: step_async / step_wait ------------------------------------
 * (env_wrappers/vectorized_env/vec_env.py:47-72; SubprocVecEnv.step_async sends the actions to its workers and returns,
 * step_wait collects: subproc_vec_env.py:50-63.)  Here the "worker" is a launch queue (HIP stream) OWNED BY THE HANDLE:
 *   scg_step_async(h, io, n, stream): everything queued on `stream` so far (the policy that produced io->action) happens
 *     before the step; the step kernel itself is issued on the handle's own queue and the call returns.  Work queued on
 *     `stream` AFTERWARDS does not wait for it.
 *   scg_step_wait(h, stream): whatever is queued on `stream` from now on happens after the handle's pending step (a device-side
 *     event wait; the host never blocks).  Without a pending step it is a no-op.
 * Why: a step kernel spends ~5 of its ~28 us (2^20 envs) ramping up and draining.  A caller that cuts its batch into two
 * (or more) independent shards -- one handle each, scg_config.env_index_base = the shard's first global env index, so the
 * Philox streams and therefore the results are those of the unsplit batch, bit for bit -- and interleaves
 *     policy(A); step_async(A); policy(B); step_async(B); step_wait(A); policy(A); ...
 * keeps two step kernels in flight: one shard's ramp-up / drain (and its policy) overlaps the other shard's body, which is
 * what scg_rollout() does for a pre-recorded action sequence -- with a consumer in the loop.  `stream` may be the handle's
 * own queue (scg_step_queue): then both calls add no event at all and step_async is scg_step on that queue.  At most one
 * step may be pending per handle; scg_step_async on a handle with a pending step orders the new step behind it (same
 * queue).  Every other entry point that launches on a caller's stream (scg_step, scg_reset, scg_rollout ...) first makes
 * that stream wait for a pending asynchronous step, so mixing the two forms is safe, only serial. */
int scg_step_async(scg_handle* h, const scg_step_io* io, int64_t n, void* stream);
int scg_step_wait(scg_handle* h, void* stream);
/* The handle's own launch queue (a hipStream_t, created on first use; owned by the handle, destroyed with it).  A caller may
 * run its policy on it (torch.cuda.ExternalStream) so that a shard's policy and step need no cross-stream event. */
int scg_step_queue(scg_handle* h, void** stream);

/* ---- a trajectory sink for the multi-step forms --------------------------------------------------------------------------
 * scg_rollout / scg_rollout_fused reuse every output buffer per step: after the call they hold the LAST step's values.  The
 * harness they stand for uses every step (tests/scripts/verbose_api.py:71-108 prints each one) and the consumer of a rollout
 * is a [T, N, ...] buffer (controllers/ppo/ppo_utils.py:308-336).  With a sink, step k writes its observation / reward /
 * done (and, when given, constraint values / violation / truncated / mse) at base + k * stride instead -- straight into
 * time-major rollout-buffer planes [T][...][ld].  No extra byte moves:
 *   - per-step launches (scg_rollout_trajectory, fused = 0): when the observation IS the state (io->obs NULL, or the shared
 *     goal-horizon buffer io->obs == io->state) the state ping-pongs THROUGH the sink -- step k reads slab k - 1 (step 0:
 *     io->state) and writes slab k (scg_step_io.state_out), so the trajectory is the chain of states itself, 8 nx + 4 nu + 13
 *     bytes per env-step as ever; the last slab is copied back to io->state once per call (device-to-device, after the last
 *     step; SCG_SINK_KEEP_STATE_IN_SINK skips it and leaves io->state stale).  An observation buffer of its own (observation
 *     noise) is simply re-pointed per step.
 *   - one launch (fused = 1; lean configurations as scg_rollout_fused, others fall back to per-step launches): each lane
 *     keeps its env in registers and streams the per-step outputs out: 4 nu bytes read + 4 nx + 5 (+ 4 nc + 1) written per
 *     env-step -- the launch-free form WITH the Gym outputs.
 * Plane strides inside a slab are io->ld (obs: obs_dim planes; constraint_values: nc planes).  Strides are in ELEMENTS of
 * the array's type.  Members left NULL keep their scg_step_io behaviour (last step's values).  terminal_record is not sunk.
 * Results per env are bit-identical to a scg_step loop. */
#define SCG_SINK_KEEP_STATE_IN_SINK 1u
typedef struct scg_trajectory_sink {
    uint32_t struct_size;            /* = sizeof(scg_trajectory_sink) */
    uint32_t flags;                  /* SCG_SINK_* */
    float* obs;                      /* [steps][obs_dim][ld]: what step() returns as obs (the reset observation for envs that
                                        finished and were auto-reset, dummy_vec_env.py:41-47) */
    int64_t obs_step_stride;         /* >= obs_dim * ld */
    float* reward;                   /* nullable [steps][>= n] */
    int64_t reward_step_stride;
    uint8_t* done;                   /* nullable [steps][>= n] */
    int64_t done_step_stride;
    float* constraint_values;        /* nullable [steps][nc][ld] */
    int64_t constraint_values_step_stride;
    uint8_t* constraint_violation;   /* nullable [steps][>= n] */
    int64_t constraint_violation_step_stride;
    uint8_t* truncated;              /* nullable [steps][>= n]; per-step launches only (info tier) */
    int64_t truncated_step_stride;
    float* mse;                      /* nullable [steps][>= n]; per-step launches only (info tier) */
    int64_t mse_step_stride;
} scg_trajectory_sink;
int scg_rollout_trajectory(scg_handle* h, const scg_step_io* io, const scg_trajectory_sink* sink, int64_t n, int32_t steps,
                           int64_t action_stride, int32_t action_period, int32_t fused, void* stream);

/* ConstraintList.get_values (constraints.py:503-520) on explicit buffers: state [nx][ld],
 * action [nu][ld] (nullable -> only_state semantics: input rows are written as NaN), out [nc][ld]. */
int scg_eval_constraints(scg_handle* h, const float* state, const float* action, float* c_val,
                         uint8_t* violation, int64_t ld, int64_t n, void* stream);

/* ---- next row: returns / GAE over a device-resident rollout ------------------
 * Replaces compute_returns_and_advantages (controllers/ppo/ppo_utils.py:428-456).  All arrays are device
 * pointers, time-major planes [T][ld] (the PPOBuffer scheme's (T, N, 1) arrays, ppo_utils.py:308-336);
 * last_val is [n].  Exactly one of mask / done: mask = the buffer's float (1 - done) planes (ppo.py:262), or done = the step
 * kernel's own byte planes [T][ld] (scg_step_io.done written straight into the rollout buffer: mask = 1 - done is formed in
 * the kernel, 1 byte instead of 4 read per transition and no conversion pass).  term_val may be NULL (no time-limit bootstrap).  rew_out (nullable) receives
 * rew + gamma * term_val, the reference's in-place update (:443).  stats (nullable, [2] doubles, zeroed by the
 * caller) accumulates sum(adv) and sum(adv^2) for the advantage normalisation of ppo.py:288. */
int scg_gae(const float* rew, const float* val, const float* mask, const uint8_t* done, const float* term_val,
            const float* last_val, float* ret, float* adv, float* rew_out, double* stats, int32_t T, int64_t n, int64_t ld,
            float gamma, float gae_lambda, int32_t use_gae, void* stream);
/* adv <- (adv - mean) / (std + eps), population statistics from `stats` (device memory, no host sync). */
int scg_normalize_advantages(float* adv, const double* stats, int32_t T, int64_t n, int64_t ld, float eps,
                             void* stream);

/* PPOBuffer.push (controllers/ppo/ppo_utils.py:338-346: `self.__dict__[k][self.t] = v` for every key of the step's batch, six or
 * seven device-to-device copies per step) as ONE launch: segment j copies a rows x cols float32 matrix, element (i, k) from
 * src[i * src_row_stride + k * src_col_stride] to dst[i * dst_row_stride + k * dst_col_stride] (strides in ELEMENTS, so a
 * row-major [N, d] tensor, an [N, d] view of SoA planes and the planes themselves are all addressable: pushing row-major actions
 * into action planes is the transposing case).  1 <= count <= SCG_MAX_COPY_SEGMENTS; the segment array is HOST memory (it
 * travels as kernel arguments); segments must not overlap each other's destinations.  Handle-free: runs on the current device. */
#define SCG_MAX_COPY_SEGMENTS 8
typedef struct scg_copy_segment {
    const float* src;
    float* dst;
    int64_t rows, cols;
    int64_t src_row_stride, src_col_stride;
    int64_t dst_row_stride, dst_col_stride;
} scg_copy_segment;
int scg_copy_segments(const scg_copy_segment* segments, int32_t count, void* stream);

/* ---- next row: batched prior-model queries ------------------------------------
 * scg_linearize replaces SymbolicModel.df_func (math_and_models/symbolic_systems.py:77-92: dfdx = d f / d x,
 * dfdu = d f / d u of the continuous dynamics) for n (x, u) pairs in one launch.  state [nx][ld] and input
 * [nu][ld] are SoA planes; `input` is the MODEL input u (per-motor thrusts in N / cart force), i.e. the
 * clipped action, not the raw action.  inertial (nullable, [4][ld]) gives per-env (M, Ixx, Iyy, Izz) /
 * (pole_length, pole_mass, cart_mass, -); NULL uses the handle's nominal constants.  Outputs are row-major
 * batches dfdx [n][nx][nx], dfdu [n][nx][nu].  disc_dt > 0 applies discretize_linear_system(exact=False)
 * (controllers/mpc/mpc_utils.py:51-54, controllers/lqr/lqr_utils.py:39-60): Ad = I + A dt, Bd = B dt; the exact=True
 * branch (:45-50) is scg_discretize_exact() below, applied to this function's outputs. */
/* SymbolicModel.fc_func (symbolic_systems.py:68) and fd_func (:70-73) for n explicit (x, u) pairs: xdot = f(x, u) and/or
 * xnext = one control step of the handle's integrator (RK4 / Euler, its sub-steps) with zero-order-hold input -- the
 * engine's discretisation, where the reference's fd_func integrates with CVODES.  Same plane conventions as scg_linearize
 * (`input` is the model input; inertial nullable).  xdot and xnext are [nx][ld] planes, each nullable, not both. */
int scg_model_eval(scg_handle* h, const float* state, const float* input, const float* inertial, float* xdot,
                   float* xnext, int64_t ld, int64_t n, void* stream);
/* SymbolicModel.fd_func AS THE REFERENCE DEFINES IT (symbolic_systems.py:70-73: cs.integrator('fd', 'cvodes', {x, p, ode},
 * {'tf': dt})): the flow of x' = f(x, p) over [0, dt] under a constant model input, in float64, to the tolerance the caller
 * states (CasADi's CVODES defaults are rtol 1e-6, atol 1e-8), for n (x0, p) pairs in one launch -- and, on request, its
 * forward sensitivities, the integrator's jac_xf_x0 / jac_xf_p.  state [nx][ld], input [nu][ld], inertial (nullable,
 * [4][ld]) and xf [nx][ld] are float64 SoA planes; dxf_dx0 [n][nx][nx] and dxf_dp [n][nx][nu] are row-major float64
 * batches; at least one of the three outputs.  Error control is step doubling on RK4 (m and 2m steps; accepted when
 * |y_2m - y_m| / 15 <= atol + rtol |y| in every component, m doubled at most max_doublings <= 16 times; the value is the
 * order-5 extrapolation); intervals (nullable, [n]) receives the 2m of the accepted solution, NEGATED where the tolerance
 * was not met (or the state is not finite).  The sensitivities are central differences of the same discretisation
 * (internal numerical differentiation: ~1e-9 relative).  dt need not be the handle's control period.  n <= 2^30
 * (2^26 with sensitivities). */
int scg_model_flow(scg_handle* h, const double* state, const double* input, const double* inertial, double dt, double rtol,
                   double atol, int32_t max_doublings, double* xf, double* dxf_dx0, double* dxf_dp, int32_t* intervals,
                   int64_t ld, int64_t n, void* stream);
int scg_linearize(scg_handle* h, const float* state, const float* input, const float* inertial, float* dfdx,
                  float* dfdu, double disc_dt, int64_t ld, int64_t n, void* stream);
/* SymbolicModel.fc_linear_func (symbolic_systems.py:93-101): the model linearised about (state, input), evaluated at
 * (x_eval, u_eval):  out = f(x, u) + dfdx(x, u) (x_eval - x) + dfdu(x, u) (u_eval - u).  All five arrays are SoA planes
 * ([nx][ld] / [nu][ld]); x_eval == NULL evaluates at x_eval = 0, which gives the constant term c of the affine ODE
 * z' = dfdx z + c that fd_linear_func (:102-107) integrates (see scg_discretize_exact).  g_linear_func (:108-112) is the
 * identity map for these systems (y = x) and needs no kernel. */
int scg_fc_linear(scg_handle* h, const float* x_eval, const float* u_eval, const float* state, const float* input,
                  const float* inertial, float* out, int64_t ld, int64_t n, void* stream);
/* discretize_linear_system(A, B, dt, exact=True) (controllers/mpc/mpc_utils.py:24-54) for n systems in one launch:
 * [[Ad, Bd], [0, I]] = expm([[A, B], [0, 0]] dt), computed on the matrix cores (Taylor + scaling and squaring on
 * v_mfma_f32_16x16x4_f32: exact f32 products).  A [n][nx][nx], B [n][nx][m] row-major device batches (the output format of
 * scg_linearize), nx + m <= 16; Ad / Bd (each nullable) have the same shapes.  Optionally the one-step prediction of
 * the discrete model: z1 [n][nx] = Ad z0 + Bd w with z0 [n][nx], w [n][m] (w == NULL: a vector of ones -- with B := the
 * constant term of an affine ODE, m = 1, this is the exact flow that SymbolicModel.fd_linear_func integrates with CVODES,
 * symbolic_systems.py:102-107).  Needs no handle: pure linear algebra. */
int scg_discretize_exact(const float* A, const float* B, int32_t nx, int32_t m, double dt, float* Ad, float* Bd,
                         const float* z0, const float* w, float* z1, int64_t n, void* stream);
/* Quadratic stage cost and gradients (quadrotor.py:678-682, symbolic_systems.py:94-123):
 * l = 1/2 ex'Q ex + 1/2 eu'R eu, l_x = 1/2 (Q + Q') ex, l_u = 1/2 (R + R') eu with ex = x - xr, eu = u - ur.
 * Q [nx*nx] and R [nu*nu] are HOST row-major doubles.  xr is a device vector [nx] when xr_ld == 0, else planes
 * [nx][xr_ld] (one reference per env); likewise ur.  Outputs (each nullable): l [n], l_x [nx][ld], l_u [nu][ld]. */
int scg_quadratic_loss(scg_handle* h, const double* Q, const double* R, const float* state, const float* input,
                       const float* xr, int64_t xr_ld, const float* ur, int64_t ur_ld, float* l, float* l_x,
                       float* l_u, int64_t ld, int64_t n, void* stream);

/* ---- next row: running mean / std normalisers ------------------------------------
 * MeanStdNormalizer.__call__ (math_and_models/normalization.py:125-133) on SoA planes x [d][ld_x] -> y [d][ld_y]:
 * update RunningMeanStd (:34-62) with the batch of n rows, then y = clip((x - mean) / sqrt(var + epsilon), +-clip).
 * state_in / state_out are device records of 2 d + 1 doubles: mean[d], var[d], count (the caller ping-pongs two
 * records; a fresh tracker is mean 0, var 1, count 1e-4, :28-30).  state_out == NULL is the read_only normaliser
 * (:126): no update.  scratch is device memory for 2 max(64 d, 1024) doubles (per-workgroup partial sums).  y may alias x.
 * y == NULL (with state_out) is RunningMeanStd.update(batch) alone (:34-46): the tracker moves, nothing is normalised. */
int scg_mean_std_normalize(const float* x, float* y, int32_t d, int64_t ld_x, int64_t ld_y, int64_t n,
                           const double* state_in, double* state_out, double* scratch, float clip, double epsilon,
                           void* stream);
/* RewardStdNormalizer.__call__ (:172-189): ret = ret * gamma + rew; update the tracker (d = 1 record) with ret;
 * clear ret of finished envs; out = clip(rew / sqrt(var + epsilon), +-clip).  scratch: 3072 doubles.
 * legacy_index_reset != 0 reproduces `self.ret[dones.astype(np.long)] = 0` literally (it indexes with the VALUES 0 / 1
 * of dones: ret[0] is cleared when some env is not done, ret[1] when some env is done); 0 clears ret where done. */
int scg_reward_normalize(const float* rew, const uint8_t* done, float* ret, float* out, int64_t n,
                         const double* state_in, double* state_out, double* scratch, float gamma, float clip,
                         double epsilon, int32_t legacy_index_reset, void* stream);

/* ---- one statistics tracker, the batch spread over several devices ------------------------------------------------
 * The reference normalises in ONE process over the observations of ALL its envs (controllers/ppo/ppo.py:260-261 behind a
 * SubprocVecEnv); with one process per GPU every rank holds a shard of the batch.  The two passes of the normalisers apart, with
 * the one exchange the path has between them: pass 1 leaves this shard's sums -- sums[j] = {sum (x_j - c_j), sum (x_j - c_j)^2},
 * c_j = the tracker's current mean (identical on every rank as long as every update goes through this path) -- the caller adds the
 * shards' sums and row counts (one all-reduce of 2 d + 1 doubles: RCCL over xGMI), pass 2 updates the tracker from the added sums
 * with batch_count = the rows of all shards and normalises this shard's rows.  Every rank ends up with the same record, the one a
 * single tracker fed with the whole batch would hold.  scratch as for scg_mean_std_normalize; sums: 2 d doubles on the device. */
int scg_mean_std_moments(const float* x, int32_t d, int64_t ld_x, int64_t n, const double* state_in, double* scratch, double* sums,
                         void* stream);
int scg_mean_std_normalize_sums(const float* x, float* y, int32_t d, int64_t ld_x, int64_t ld_y, int64_t n, const double* state_in,
                                double* state_out, const double* sums, double batch_count, float clip, double epsilon, void* stream);
/* The same split of RewardStdNormalizer (the returns of finished envs are cleared, i.e. legacy_index_reset = 0): pass 1 advances
 * ret and leaves sums[3] = {sum (ret - c), sum (ret - c)^2, finished envs}; pass 2 takes the added sums. */
int scg_reward_moments(const float* rew, const uint8_t* done, float* ret, int64_t n, const double* state_in, double* scratch,
                       double* sums, float gamma, void* stream);
int scg_reward_normalize_sums(const float* rew, float* out, int64_t n, const double* state_in, double* state_out, const double* sums,
                              double batch_count, float clip, double epsilon, void* stream);

/* Introspection of scg_rollout's hipGraph replay (tests, tuning): *cached = rollout signatures currently held as instantiated
 * graphs (-1: a capture failed and the handle stays on plain launches), *replays = graph launches so far.  Either may be NULL. */
int scg_rollout_graph_info(const scg_handle* h, int32_t* cached, int64_t* replays);

/* Introspection */
int scg_num_constraints(const scg_handle* h);
int scg_obs_dim(const scg_handle* h);
int scg_kernel_info(const scg_handle* h, int32_t* block, int32_t* grid_for_n, int32_t* lds_bytes, int64_t n);
const char* scg_last_error(void);
int scg_version(void);
/* sizeof() of the ABI structs as compiled, so a binding can verify its own layout */
int scg_sizeof_config(void);
int scg_sizeof_step_io(void);
int scg_sizeof_constraint_row(void);
int scg_sizeof_disturbance(void);
int scg_sizeof_trajectory_sink(void);

#ifdef __cplusplus
}
#endif
#endif /* SCG_ROLLOUT_H */
