/* Plain-C float64 restatement of the step path, batched with OpenMP.  TEST INFRASTRUCTURE ONLY
 * (see oracle/__init__.py): used by tests/ (validated against the Python oracle) and by the
 * cpu_baseline leg of bench.py.  Layout is array-of-structs, row-major [env][component], like
 * the reference's per-env NumPy vectors.
 *
 * Reference anchors (relative to /root/reference/safe_control_gym):
 *   envs/gym_pybullet_drones/quadrotor.py:507-558 (step order), :594-676 (dynamics),
 *   :815-842 + quadrotor_utils.py:21-67 (action preprocessing), :886-930 (reward),
 *   :956-993 (done), benchmark_env.py:422-463 (after_step), controllers/mpc/mpc_utils.py:79-101 (RK4),
 *   envs/constraints.py:273-304 (bounded rows), env_wrappers/vectorized_env/dummy_vec_env.py:41-47
 *   (auto-reset).  CartPole: DESIGN.md row A16 (not in the reference fork). */
#ifndef SCG_ORACLE_H
#define SCG_ORACLE_H
#include <stdint.h>

typedef struct scgo_config {
    int32_t system;          /* 1,2,3 quadrotor; 4 cartpole */
    int32_t euler;           /* 0 = RK4, 1 = explicit Euler */
    int32_t substeps;
    int32_t tracking;        /* Task.TRAJ_TRACKING */
    int32_t quadratic;       /* Cost.QUADRATIC (diagonal Q,R in w_state/w_act), else rl_reward */
    int32_t rew_exponential;
    int32_t ctrl_steps;
    int32_t motor_map;
    int32_t done_on_oob;
    int32_t done_on_violation;
    int32_t n_bounds;        /* bounded-constraint rows: c = sign * v[dim] - b, v = [state; raw action] */
    int32_t ref_rows;
    double dt;
    double m, g, l, ixx, iyy, izz, kf, km, pwm_scale, pwm_const, pwm_min, pwm_max;
    double cp_l, cp_m, cp_M;
    double act_low[4], act_high[4], u_goal[4];
    double x_goal[12], w_state[12], w_act[4];
    double oob_low[12], oob_high[12];
    double goal_tol;
    int32_t bound_dim[32];
    double bound_sign[32], bound_b[32];
} scgo_config;

/* One step() for n envs.  action [n][nu] (raw), state [n][nx] in/out, t [n] in/out, ref [ref_rows][nx],
 * reset_state [n][nx] (state after an auto-reset), c_val [n][n_bounds] (nullable), viol [n] (nullable). */
void scgo_step(const scgo_config* c, int64_t n, const double* action, double* state, int32_t* t,
               double* reward, uint8_t* done, const double* ref, const double* reset_state, double* c_val,
               uint8_t* viol, int32_t num_threads);
int scgo_max_threads(void);
#endif
