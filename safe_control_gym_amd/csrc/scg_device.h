// Internal device-side parameter blocks of the rollout engine (not part of the C ABI).
// Everything here is float: the kernel computes in fp32; the host forms derived constants in
// double (scg_capi.hip) and rounds once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/scg_rollout.h"

namespace scg {

enum Flags : uint32_t {
    F_NORMALIZED = 1u << 0,     // NORMALIZED_RL_ACTION_SPACE
    F_MOTOR_MAP = 1u << 1,      // cmd2pwm -> pwm2rpm -> KF rpm^2 round trip
    F_REW_EXP = 1u << 2,        // rew_exponential
    F_Q_DIAG = 1u << 3,         // Q and R are diagonal (fast path)
    F_DONE_OOB = 1u << 4,       // done_on_out_of_bound
    F_DONE_VIOL = 1u << 5,      // done_on_violation
    F_PENALTY = 1u << 6,        // use_constraint_penalty (rl_reward only)
    F_RAND_INIT = 1u << 7,      // randomized_init
    F_AUTO_RESET = 1u << 8,     // fused DummyVecEnv auto-reset
    F_TRACKING = 1u << 9,       // Task.TRAJ_TRACKING
    F_QUADRATIC = 1u << 10,     // Cost.QUADRATIC (else RL_REWARD)
    F_EULER = 1u << 11,         // explicit Euler instead of RK4
    F_REF_IN_LDS = 1u << 12,    // reference table staged in LDS (else read through L1/L2)
    F_RAND_INERTIAL = 1u << 13, // per-env inertial properties re-drawn at every reset
    F_GND = 1u << 14,           // ground effect (Physics.PYB_GND)
    F_DRAG = 1u << 15,          // drag (Physics.PYB_DRAG)
    F_W_STATE_UNIT = 1u << 17,  // every diagonal state weight of the cost is exactly 1 (Q = I / rew_state_weight = 1)
    F_RAND_GENERAL = 1u << 16,  // some reset-time quantity is drawn from a non-uniform distribution (scg_distribution)
    F_LONE = 1u << 18,          // launch policy, set per launch by scg_step / scg_step_async: this kernel runs alone on its queue with
                                // nothing of its own kind overlapping it (a policy sits between two steps).  Same results bit for bit;
                                // it selects the forms that measured faster for a lone launch (profiles/r04/lone_launch_arms.txt): the
                                // per-wave cooperative reset instead of the workgroup-pooled one (two workgroup barriers per step),
                                // plain instead of nontemporal terminal-record stores
    F_SERIAL_WORD = 1u << 19,   // serial_base is set: add *serial_base to the reset serial (launches replayed from a rollout graph)
};

// Physical constants of the model, in the real type the integrator runs in: float in the kernarg block for
// the default path, double in KTables for the precise (fp64-integration) mode -- there a 1e-8 mismatch of
// a constant is a model bias that the open-loop plant integrates four times (torque -> position ~ t^4).
template <class T>
struct PhysT {
    T h;                        // dt / substeps
    // quadrotor
    T inv_m, g;
    T arm_s2;                   // l / sqrt(2)
    T gamma;                    // KM / KF
    T inv_ixx, inv_iyy, inv_izz;
    T cxx, cyy, czz;            // (Izz-Iyy)/Ixx, (Ixx-Izz)/Iyy, (Iyy-Ixx)/Izz
    T l_s2_iyy;                 // l / sqrt(2) / Iyy  (2-D)
    // motor map
    T inv_nmotor_kf;            // 1 / (n_motor KF)
    T nmotor_kf;                // n_motor KF
    T pwm_scale, inv_pwm_scale, pwm_const, pwm_min, pwm_max;
    T th_min, th_max;           // n KF (pwm_scale pwm_min + pwm_const)^2, ... pwm_max ...: the pwm clip seen from the thrust side
    // cartpole
    T cp_l, cp_m, cp_inv_Mm, cp_ml, cp_g;
};

struct DistTable;

// Scalars: travel in the kernarg block (SGPRs).
struct KParams {
    uint32_t flags;
    int32_t substeps;
    int32_t ctrl_steps;
    int32_t ref_rows;           // rows of the X_GOAL table (tracking)
    int32_t obs_goal_horizon;
    int32_t n_con_rows;         // total constraint rows (c_val planes)
    int32_t grid_cap_per_cu;    // host-side launch policy (scg_config.grid_cap_per_cu); not read on the device
    uint32_t oob_dims;          // bit j: state dim j has a binding out-of-bound limit (|bound| < f32max); the others are skipped
    uint32_t oob_sym;           // bit j: that limit is symmetric (low == -high in float32): |x| <= high is the whole test
    uint32_t seed_lo, seed_hi;
    int64_t env_base;
    PhysT<float> ph;            // physical constants (fp32 copy, SGPRs)
    // action
    float act_mul, act_add;     // preprocessed action = act_mul * clip(a) + act_add: (1, 0), or the normalised action space
                                // (quadrotor.py:828: (norm_act_scale hover_thrust, hover_thrust); cart-pole: (action_scale, 0))
    float goal_tol2;
    float constraint_penalty;
    float cost_scale;           // -1 (rl_reward: rew = -dist) or -0.5 (quadratic: rew = -l/2 ... see w_state)
    // aero terms: gnd_k = gnd_eff_coeff (prop_radius / 4)^2, drag_* = drag_coeff * 2 pi / 60 (per rpm)
    float gnd_k, gnd_h_clip, drag_xy, drag_z;
    const DistTable* dist;   // device; null when the env has no passive disturbances
    const uint32_t* serial_base;   // device word added to scg_step_io.reset_serial (launches replayed from a rollout graph); null otherwise
};

// Per-dimension constant vectors: live in device memory, read through a wave-uniform pointer with
// scalar loads right where they are used (keeping them out of the kernarg block keeps them out of
// the SGPR file for the whole kernel -- 100+ SGPRs of pressure otherwise).
struct KTables {
    float act_low[SCG_MAX_NU], act_high[SCG_MAX_NU];
    float u_goal[SCG_MAX_NU];
    float x_goal[SCG_MAX_NX];
    // diagonal cost weights: rl_reward -> rew_state_weight / rew_act_weight (quadrotor.py:902-908);
    // quadratic with diagonal Q, R -> diag(Q) / diag(R) (quadrotor.py:682)
    float w_state[SCG_MAX_NX], w_act[SCG_MAX_NU];
    float mse_w[SCG_MAX_NX];
    float oob_low[SCG_MAX_NX], oob_high[SCG_MAX_NX];
    float init_state[SCG_MAX_NX], init_lo[SCG_MAX_NX], init_span[SCG_MAX_NX];
    PhysT<double> phd;          // double-precision physical constants (precise mode only)
    float inertial_nominal[4], inertial_lo[4], inertial_span[4];   // (M, Ixx, Iyy, Izz) | (pole_length, pole_mass, cart_mass, -)
    double arm, kmkf;           // l and KM/KF, to rebuild the per-env constants
    double act_low_d[SCG_MAX_NU], act_high_d[SCG_MAX_NU];
    // distribution of every randomised quantity (only read when F_RAND_GENERAL is set; uniform dims keep lo / span above)
    int32_t init_kind[SCG_MAX_NX], inertial_kind[4];
    float init_p[SCG_MAX_NX][3], inertial_p[4][3];
    // full Q / R (only read when F_Q_DIAG is clear)
    float Q[SCG_MAX_NX * SCG_MAX_NX];
    float R[SCG_MAX_NU * SCG_MAX_NU];
};

enum RowFlags : int32_t { RF_STRICT = 1, RF_HAS_TOL = 2 };

struct BoundEntry {            // c = sign * v[dim] - b   (dim is implied by the grouping below)
    float sign, b, tol;
    int32_t out_row, flags;
};
struct DenseRow {              // c = coef . v - b
    float coef[SCG_MAX_NV];
    float b, tol;
    int32_t out_row, flags;
};
struct QuadRow {               // c = v' P v - b
    float P[SCG_MAX_NV * SCG_MAX_NV];
    float b, tol;
    int32_t out_row, flags;
};

// Compiled ConstraintList.  Bound entries (rows of +-I: BoundedConstraint / DefaultConstraint)
// are grouped by the v-dimension they read, so the kernel indexes its register-resident v[]
// with compile-time indices only.
struct ConTable {
    int32_t n_rows, n_bound, n_dense, n_quad;
    int32_t n_state_dims;                       // nx: v[k] with k >= nx is the raw action
    int32_t bound_start[SCG_MAX_NV + 1];
    BoundEntry bound[SCG_MAX_CONSTRAINT_ROWS];
    DenseRow dense[SCG_MAX_CONSTRAINT_ROWS];
    QuadRow quad[SCG_MAX_QUADRATIC];
};

// Compiled DisturbanceList entries (scg_set_disturbances), grouped by mode.
struct DistEntry {
    int32_t kind, step_offset, max_step, duration;
    float magnitude, log2_decay, half_duration;
    float scale, omega_dt;                      // periodic: 2 pi frequency * CTRL_TIMESTEP
    float a[SCG_MAX_NX], b[SCG_MAX_NX];         // uniform: low, high - low; white_noise: std, -
    float mask[SCG_MAX_NX];                     // 1 when the entry has no mask
};
struct DistTable {
    int32_t n[3];
    DistEntry e[3][SCG_MAX_DISTURBANCES];
};

struct LaunchGeom {
    int32_t block, grid, lds_bytes;
};

// Arguments of the one-launch multi-step form (scg_rollout_fused / scg_rollout_trajectory with fused = 1): step k reads its
// action planes at io.action + (k % period) * stride; the sink_* members (scg_trajectory_sink; each nullable) receive step k's
// outputs at base + k * step stride.  Single-step launches pass {1, 1, 0}.
struct RolloutArgs {
    int32_t steps, period;
    int64_t stride;
    float* sink_obs;
    int64_t sink_obs_stride;
    float* sink_rew;
    int64_t sink_rew_stride;
    uint8_t* sink_done;
    int64_t sink_done_stride;
    float* sink_cval;
    int64_t sink_cval_stride;
    uint8_t* sink_viol;
    int64_t sink_viol_stride;
};

}  // namespace scg

// Kernel launchers (scg_kernels.hip)
namespace scg {
LaunchGeom step_geometry(const KParams& p, int system, int64_t n, int num_cus);
hipError_t launch_step(int system, const KParams& p, const scg_step_io& io, const KTables* tb,
                       const ConTable* ct, const float* ref, int64_t n, int num_cus, hipStream_t s);
bool fused_rollout_ok(const KParams& p, const scg_step_io& io);
hipError_t launch_rollout_fused(int system, const KParams& p, const scg_step_io& io, const KTables* tb,
                                const ConTable* ct, const float* ref, int64_t n, const RolloutArgs& ro, int num_cus,
                                hipStream_t s);
hipError_t launch_reset(int system, const KParams& p, const KTables* tb, float* state, float* state_lo,
                        int32_t* step_count, float* ep_return, int32_t* ep_length, float* obs, const uint8_t* mask,
                        const float* reset_state, float* inertial, uint32_t* episode_id, const float* ref,
                        uint32_t reset_serial, int64_t ld, int64_t n, hipStream_t s);
hipError_t launch_gae(const float* rew, const float* val, const float* mask, const uint8_t* done, const float* term_val,
                      const float* last_val, float* ret, float* adv, float* rew_out, double* stats, int T, int64_t n,
                      int64_t ld, float gamma, float lam, int use_gae, hipStream_t s);
hipError_t launch_normalize(float* adv, const double* stats, int T, int64_t n, int64_t ld, float eps, hipStream_t s);
hipError_t launch_copy_segments(const scg_copy_segment* seg, int count, hipStream_t s);
hipError_t launch_model_eval(int system, const KParams& p, const KTables* tb, const float* state, const float* input,
                             const float* inertial, float* xdot, float* xnext, int64_t ld, int64_t n, hipStream_t s);
hipError_t launch_model_flow(int system, const PhysT<double>& ph, const double* state, const double* input,
                             const double* inertial, double dt, double rtol, double atol, int max_doublings, double* xf,
                             double* dxf_dx0, double* dxf_dp, int32_t* intervals, int64_t ld, int64_t n, hipStream_t s);
hipError_t launch_linearize(int system, const PhysT<float>& ph, const float* state, const float* input,
                            const float* inertial, float* dfdx, float* dfdu, float disc_dt, int64_t ld, int64_t n,
                            hipStream_t s);
hipError_t launch_fc_linear(int system, const PhysT<float>& ph, const float* x_eval, const float* u_eval, const float* state,
                            const float* input, const float* inertial, float* out, int64_t ld, int64_t n, hipStream_t s);
hipError_t launch_discretize_exact(const float* A, const float* B, int nx, int m, float dt, float* Ad, float* Bd,
                                   const float* z0, const float* w, float* z1, int64_t n, int num_cus, hipStream_t s);
hipError_t launch_loss(int system, const double* Q, const double* R, const float* state, const float* input,
                       const float* xr, int64_t xr_ld, const float* ur, int64_t ur_ld, float* l, float* l_x, float* l_u,
                       int64_t ld, int64_t n, hipStream_t s);
hipError_t launch_mean_std_normalize(const float* x, float* y, int d, int64_t ld_x, int64_t ld_y, int64_t n,
                                     const double* state_in, double* state_out, double* moments, float clip,
                                     double epsilon, hipStream_t s);
hipError_t launch_reward_normalize(const float* rew, const uint8_t* done, float* ret, float* out, int64_t n,
                                   const double* state_in, double* state_out, double* moments, float gamma,
                                   float clip, double epsilon, int legacy, hipStream_t s);
hipError_t launch_mean_std_moments(const float* x, int d, int64_t ld_x, int64_t n, const double* state_in, double* partials,
                                   double* sums, hipStream_t s);
hipError_t launch_mean_std_apply_sums(const float* x, float* y, int d, int64_t ld_x, int64_t ld_y, int64_t n, const double* state_in,
                                      double* state_out, const double* sums, double batch_n, float clip, double epsilon, hipStream_t s);
hipError_t launch_reward_moments(const float* rew, const uint8_t* done, float* ret, int64_t n, const double* state_in, double* partials,
                                 double* sums, float gamma, hipStream_t s);
hipError_t launch_reward_apply_sums(const float* rew, float* out, int64_t n, const double* state_in, double* state_out,
                                    const double* sums, double batch_n, float clip, double epsilon, hipStream_t s);
hipError_t launch_eval_constraints(int system, const ConTable* ct, const float* state, const float* action,
                                   float* c_val, uint8_t* violation, int64_t ld, int64_t n, hipStream_t s);
}  // namespace scg
