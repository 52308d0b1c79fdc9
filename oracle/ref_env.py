"""Oracle: single-environment ``reset()`` / ``step()`` in float64, same order of operations as the
reference, with the PyBullet sub-step loop replaced by the reference's own RK4 of its own
CasADi-derived prior model (that pair is the parity spec, not PyBullet).

Test infrastructure -- see ``oracle/__init__.py``.

Reference anchors (relative to /root/reference/safe_control_gym):
  * envs/gym_pybullet_drones/quadrotor.py:148-329   constructor, goals
  * envs/gym_pybullet_drones/quadrotor.py:356-504   reset (gates/obstacles/PyBullet parts omitted)
  * envs/gym_pybullet_drones/quadrotor.py:507-558   step: order preprocess -> force -> advance ->
                                                     obs -> info -> done -> reward -> after_step
  * envs/gym_pybullet_drones/quadrotor.py:699-806   action / state / observation spaces
  * envs/gym_pybullet_drones/quadrotor.py:815-842   _preprocess_control
  * envs/gym_pybullet_drones/quadrotor.py:844-884   _get_observation
  * envs/gym_pybullet_drones/quadrotor.py:886-930   _get_reward (rl_reward, quadratic)
  * envs/gym_pybullet_drones/quadrotor.py:956-993   _get_done
  * envs/gym_pybullet_drones/quadrotor.py:1011-1025 _get_info (mse)
  * envs/benchmark_env.py:63-218                    base constructor, timing constants, Q = R = I
  * envs/benchmark_env.py:277-308                   _randomize_values_by_info
  * envs/benchmark_env.py:356-463                   before_reset / after_reset / before_step /
                                                     extend_obs / after_step
  * math_and_models/symbolic_systems.py:114-123     loss = cost_func  (quadrotor.py:682)

Documented deviations (also in DESIGN.md):
  * ``goal_reached`` in ``info``: the reference reads ``self.goal_reached`` in ``_get_info``
    (:1013) *before* ``_get_done`` assigns it (:965) -- an AttributeError on the first step of a
    stabilisation+quadratic run.  The oracle reports this step's value.
  * RNG: the reference draws from ``gym.utils.seeding.np_random(seed)`` (gym absent here).  The
    oracle takes an explicit ``reset_sampler(env_index, episode_index) -> offsets`` so tests can
    feed it the same counter-based draws the device engine uses.
  * CartPole is not in the reference fork -- parity unpinned, spec in DESIGN.md.
"""
import math
from copy import deepcopy

import numpy as np

from . import ref_model as rm
from .ref_trajectory import build_x_goal, DEFAULT_TASK_INFO
from .ref_constraints import create_ref_constraint_list

F32MAX = float(np.finfo(np.float32).max)


def _f32(x):
    """gym.spaces.Box(dtype=float32) stores bounds as float32 (quadrotor.py:731-733,791)."""
    return np.asarray(x, dtype=np.float32).astype(np.float64)


INIT_STATE_LABELS = {                                           # quadrotor.py:207-212
    1: ['init_x', 'init_x_dot'],
    2: ['init_x', 'init_x_dot', 'init_z', 'init_z_dot', 'init_theta', 'init_theta_dot'],
    3: ['init_x', 'init_x_dot', 'init_y', 'init_y_dot', 'init_z', 'init_z_dot',
        'init_phi', 'init_theta', 'init_psi', 'init_p', 'init_q', 'init_r'],
    'cartpole': ['init_x', 'init_x_dot', 'init_theta', 'init_theta_dot'],
}

_U = lambda lo, hi: {'distrib': 'uniform', 'low': lo, 'high': hi}
QUAD_INIT_STATE_RAND_INFO = {                                   # quadrotor.py:68-134
    'init_x': _U(-0.5, 0.5), 'init_x_dot': _U(-0.01, 0.01),
    'init_y': _U(-0.5, 0.5), 'init_y_dot': _U(-0.01, 0.01),
    'init_z': _U(0.1, 1.5), 'init_z_dot': _U(-0.01, 0.01),
    'init_phi': _U(-0.3, 0.3), 'init_theta': _U(-0.3, 0.3), 'init_psi': _U(-0.3, 0.3),
    'init_p': _U(-0.01, 0.01), 'init_theta_dot': _U(-0.01, 0.01),
    'init_q': _U(-0.01, 0.01), 'init_r': _U(-0.01, 0.01),
}
CARTPOLE_INIT_STATE_RAND_INFO = {                               # tests/scripts/verbose_api.yaml:15-31
    'init_x': _U(-0.05, 0.05), 'init_x_dot': _U(-0.05, 0.05),
    'init_theta': _U(-0.05, 0.05), 'init_theta_dot': _U(-0.05, 0.05),
}


class RefEnv:
    """One environment, float64.  ``system`` is 1, 2, 3 (QuadType) or 'cartpole'."""

    def __init__(self, system=2,
                 # BenchmarkEnv args (benchmark_env.py:63-97)
                 task='stabilization', task_info=None, cost='rl_reward',
                 pyb_freq=50, ctrl_freq=50, episode_len_sec=5,
                 init_state=None, randomized_init=True, init_state_randomization_info=None,
                 inertial_prop=None,
                 constraints=None, done_on_violation=False, use_constraint_penalty=False,
                 constraint_penalty=-1, normalized_rl_action_space=False,
                 # Quadrotor args (quadrotor.py:148-161)
                 norm_act_scale=0.1, obs_goal_horizon=0, rew_state_weight=1.0, rew_act_weight=0.0001,
                 rew_exponential=True, done_on_out_of_bound=True, info_mse_metric_state_weight=None,
                 # analytic-engine args
                 integrator='rk4', substeps=1, motor_map=True, ground_effect=False, drag=False,
                 # oracle-only plumbing
                 reset_sampler=None, env_index=0, inertial_sampler=None):
        self.system = system
        self.NAME = 'cartpole' if system == 'cartpole' else 'quadrotor'
        self.nx, self.nu = rm.NX[system], rm.NU[system]
        self.state_dim, self.action_dim = self.nx, self.nu
        self.TASK, self.COST = str(task), str(cost)
        self.TASK_INFO = deepcopy(DEFAULT_TASK_INFO) if task_info is None else task_info
        if system == 'cartpole' and task_info is None:
            self.TASK_INFO = {'stabilization_goal': [0], 'stabilization_goal_tolerance': 0.05,
                              'trajectory_type': 'circle', 'num_cycles': 1, 'trajectory_plane': 'zx',
                              'trajectory_position_offset': [0, 0], 'trajectory_scale': 0.2}
        # benchmark_env.py:151-161
        self.CTRL_FREQ, self.PYB_FREQ = ctrl_freq, pyb_freq
        if pyb_freq % ctrl_freq != 0:
            raise ValueError('[ERROR] in BenchmarkEnv.__init__(), pyb_freq is not divisible by env_freq.')
        self.PYB_STEPS_PER_CTRL = int(pyb_freq / ctrl_freq)
        self.CTRL_TIMESTEP = 1. / ctrl_freq
        self.EPISODE_LEN_SEC = episode_len_sec
        self.CTRL_STEPS = episode_len_sec * ctrl_freq
        self.RANDOMIZED_INIT = randomized_init
        self.NORMALIZED_RL_ACTION_SPACE = normalized_rl_action_space
        self.norm_act_scale = norm_act_scale
        self.obs_goal_horizon = obs_goal_horizon
        self.rew_state_weight = np.array(rew_state_weight, ndmin=1, dtype=float)
        self.rew_act_weight = np.array(rew_act_weight, ndmin=1, dtype=float)
        self.rew_exponential = rew_exponential
        self.done_on_out_of_bound = done_on_out_of_bound
        self.integrator, self.substeps, self.use_motor_map = integrator, int(substeps), motor_map
        self.ground_effect, self.drag = bool(ground_effect), bool(drag)     # Physics.PYB_GND / PYB_DRAG analogues
        self.reset_sampler, self.env_index = reset_sampler, env_index
        self.inertial_sampler = inertial_sampler      # (env_index, episode) -> offsets added to (M, Ixx, Iyy, Izz)
        self.episode_index = -1
        # quadrotor.py:185-200
        if info_mse_metric_state_weight is None:
            w = {1: [1, 0], 2: [1, 0, 1, 0, 0, 0], 3: [1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0],
                 'cartpole': [1, 0, 1, 0]}[system]
        else:
            if len(info_mse_metric_state_weight) != self.nx:
                raise ValueError('[ERROR] wrong info_mse_metric_state_weight argument size.')
            w = info_mse_metric_state_weight
        self.info_mse_metric_state_weight = np.array(w, ndmin=1, dtype=float)
        # physical parameters (quadrotor.py:242-256)
        if system == 'cartpole':
            ip = inertial_prop or {}
            self.params = rm.CartPoleParams(ip.get('pole_length', 0.5), ip.get('pole_mass', 0.1),
                                            ip.get('cart_mass', 1.0))
        else:
            self.params = rm.QuadParams()
            if inertial_prop is None:
                pass
            elif isinstance(inertial_prop, dict):
                self.params = rm.QuadParams(inertial_prop.get('M'), inertial_prop.get('Ixx'),
                                            inertial_prop.get('Iyy'), inertial_prop.get('Izz'))
            elif system == 1 and np.array(inertial_prop).shape == (1,):
                self.params = rm.QuadParams(m=inertial_prop[0])
            elif system == 2 and np.array(inertial_prop).shape == (2,):
                self.params = rm.QuadParams(m=inertial_prop[0], iyy=inertial_prop[1])
            elif system == 3 and np.array(inertial_prop).shape == (4,):
                self.params = rm.QuadParams(*inertial_prop)
            else:
                raise ValueError('[ERROR] in Quadrotor.__init__(), inertial_prop incorrect format.')
        self._set_action_space()
        self._set_observation_space()
        # benchmark_env.py:190-192
        if self.COST == 'quadratic':
            self.Q = np.eye(self.obs_dim)
            self.R = np.eye(self.nu)
        # initial state (quadrotor.py:213-224)
        labels = INIT_STATE_LABELS[system]
        if init_state is None:
            self.init_nominal = np.zeros(self.nx)
        elif isinstance(init_state, np.ndarray):
            self.init_nominal = np.array(init_state, dtype=float)
        elif isinstance(init_state, dict):
            self.init_nominal = np.array([init_state.get(k, 0.) for k in labels], dtype=float)
        else:
            raise ValueError('[ERROR] in Quadrotor.__init__(), init_state incorrect format.')
        rand_info = init_state_randomization_info
        if rand_info is None:
            rand_info = CARTPOLE_INIT_STATE_RAND_INFO if system == 'cartpole' else QUAD_INIT_STATE_RAND_INFO
        self.INIT_STATE_RAND_INFO = {k: v for k, v in rand_info.items() if k in labels}  # :227-229
        # goals (quadrotor.py:261-329)
        if system == 'cartpole':
            self.U_GOAL = np.zeros(1)
            if self.TASK == 'stabilization':
                self.X_GOAL = np.hstack([self.TASK_INFO['stabilization_goal'][0], 0., 0., 0.])
            else:
                from .ref_trajectory import generate_trajectory
                ti = self.TASK_INFO
                pos, vel, _ = generate_trajectory(ti['trajectory_type'], self.EPISODE_LEN_SEC,
                                                  ti['num_cycles'], ti['trajectory_plane'],
                                                  ti['trajectory_position_offset'],
                                                  ti['trajectory_scale'], self.CTRL_TIMESTEP)
                z = np.zeros(pos.shape[0])
                self.X_GOAL = np.vstack([pos[:, 0], vel[:, 0], z, z]).transpose()
        else:
            self.U_GOAL = np.ones(self.nu) * self.params.m * self.params.g / self.nu
            self.X_GOAL = build_x_goal(system, self.TASK, self.TASK_INFO, self.EPISODE_LEN_SEC,
                                       self.CTRL_TIMESTEP)
        self.X_EQ = np.zeros(self.nx)
        self.U_EQ = self.U_GOAL
        # constraints (benchmark_env.py:194-198,332-338)
        self.DONE_ON_VIOLATION = done_on_violation
        self.use_constraint_penalty = use_constraint_penalty
        self.constraint_penalty = constraint_penalty
        self.constraints = None
        self.num_constraints = 0
        if constraints is not None:
            self.constraints = create_ref_constraint_list(constraints, self)
            self.num_constraints = self.constraints.num_constraints
        self.initial_reset = False
        self.goal_reached = False
        # optional per-step additive inputs (disturbance hooks; A15), set by the caller
        self.action_disturbance = None      # (nu,) added after clip, before motor map (quadrotor.py:835-838)
        self.dynamics_force = None          # world-frame (3,) force (quadrotor.py:526-548)
        self.obs_noise = None               # (nx,) added to obs (quadrotor.py:879-880)
        self.disturbance_model = None       # oracle.ref_disturbances.RefDisturbanceModel: fills the three hooks

    # ------------------------------------------------------------------ spaces
    def _set_action_space(self):
        if self.system == 'cartpole':
            self.action_scale = 10.0
            th = 1.0 if self.NORMALIZED_RL_ACTION_SPACE else self.action_scale
            self.action_space_low = _f32(np.full(1, -th))
            self.action_space_high = _f32(np.full(1, th))
            return
        if self.NORMALIZED_RL_ACTION_SPACE:                     # quadrotor.py:720-725
            self.hover_thrust = self.params.g * self.params.m / self.nu
            self.action_space_low = _f32(-np.ones(self.nu))
            self.action_space_high = _f32(np.ones(self.nu))
        else:                                                   # :726-733
            lo, hi = rm.action_bounds(self.nu, self.params)
            self.action_space_low = np.full(self.nu, lo)
            self.action_space_high = np.full(self.nu, hi)

    def _set_observation_space(self):
        r85 = 85 * math.pi / 180
        r180 = 180 * math.pi / 180
        if self.system == 1:                                    # quadrotor.py:750-755
            low = [0, -F32MAX]
            high = [2.5, F32MAX]
        elif self.system == 2:                                  # :756-769
            low = [-5, -F32MAX, 0, -F32MAX, -r85, -F32MAX]
            high = [5, F32MAX, 2.5, F32MAX, r85, F32MAX]
        elif self.system == 3:                                  # :770-789
            low = [-5, -F32MAX, -5, -F32MAX, 0, -F32MAX, -r85, -r85, -r180, -F32MAX, -F32MAX, -F32MAX]
            high = [5, F32MAX, 5, F32MAX, 2.5, F32MAX, r85, r85, r180, F32MAX, F32MAX, F32MAX]
        else:                                                   # upstream cartpole: x 2.4, theta 90 deg, doubled
            self.x_threshold, self.theta_threshold_radians = 2.4, 90 * math.pi / 180
            low = [-self.x_threshold * 2, -F32MAX, -self.theta_threshold_radians * 2, -F32MAX]
            high = [self.x_threshold * 2, F32MAX, self.theta_threshold_radians * 2, F32MAX]
        self.state_space_low, self.state_space_high = _f32(low), _f32(high)
        mul = 1
        if self.COST == 'rl_reward' and self.obs_goal_horizon > 0:   # :794-802
            mul = 1 + self.obs_goal_horizon if self.TASK == 'traj_tracking' else 2
        self.obs_dim = self.nx * mul

    # ------------------------------------------------------------------ reset
    def reset(self):
        """quadrotor.py:356-504 + benchmark_env.py:356-383."""
        self.initial_reset = True
        self.ctrl_step_counter = 0
        self.pyb_step_counter = 0
        self.current_raw_input_action = None
        self.current_preprocessed_action = None
        self.episode_index += 1
        if self.inertial_sampler is not None:                   # quadrotor.py:455-467: nominal + draw, every reset
            if not hasattr(self, '_nominal_params'):
                self._nominal_params = self.params
            off = self.inertial_sampler(self.env_index, self.episode_index)
            n = self._nominal_params
            if self.system == 'cartpole':
                self.params = rm.CartPoleParams(n.l + off[0], n.m + off[1], n.M + off[2])
            else:
                self.params = rm.QuadParams(n.m + off[0], n.ixx + off[1], n.iyy + off[2], n.izz + off[3])
        init = self.init_nominal.copy()
        if self.RANDOMIZED_INIT:                                # benchmark_env.py:297-307 (adds to nominal)
            offs = self.reset_sampler(self.env_index, self.episode_index)
            for j, label in enumerate(INIT_STATE_LABELS[self.system]):
                if label in self.INIT_STATE_RAND_INFO:
                    init[j] += offs[j]
        self.state = init
        if self.disturbance_model is not None:
            self.disturbance_model.on_reset(self)
        obs = self._get_observation()
        info = {'x_reference': self.X_GOAL, 'u_reference': self.U_GOAL}
        if self.constraints is not None:                        # benchmark_env.py:381-382
            info['constraint_values'] = self.constraints.get_values(self, only_state=True)
        return obs, info

    # ------------------------------------------------------------------ step
    def step(self, action):
        if self.disturbance_model is not None:
            self.disturbance_model.on_step(self)
        thrust = self.before_step(action)
        fext = (0.0, 0.0, 0.0)
        if self.dynamics_force is not None:
            fext = tuple(float(v) for v in self.dynamics_force)
        self._advance_simulation(thrust, fext)
        obs = self._get_observation()
        info = self._get_info()
        done = self._get_done()
        if self.TASK == 'stabilization' and self.COST == 'quadratic':
            info['goal_reached'] = self.goal_reached
        rew = self._get_reward()
        return self.after_step(obs, rew, done, info)

    def before_step(self, action):
        """benchmark_env.py:394-404."""
        if not self.initial_reset:
            raise RuntimeError('[ERROR] You must call env.reset() at least once before using env.step().')
        self.current_raw_input_action = np.asarray(action, dtype=float)
        return self._preprocess_control(self.current_raw_input_action)

    def _preprocess_control(self, action):
        """quadrotor.py:815-842 (cartpole: clip, scale by action_scale when normalised)."""
        lo, hi = self.action_space_low, self.action_space_high
        if self.system == 'cartpole':
            force = np.clip(action, lo, hi)
            if self.NORMALIZED_RL_ACTION_SPACE:
                force = self.action_scale * force
            self.current_preprocessed_action = force
            if self.action_disturbance is not None:
                force = force + self.action_disturbance
            return list(force)
        if self.NORMALIZED_RL_ACTION_SPACE:
            a = np.clip(action, lo, hi)
            thrust = (1 + self.norm_act_scale * a) * self.hover_thrust
        else:
            thrust = np.clip(action, lo, hi)
        self.current_preprocessed_action = thrust
        if self.action_disturbance is not None:
            thrust = thrust + self.action_disturbance
        if self.use_motor_map:
            return rm.motor_map(list(thrust), self.params)
        return list(thrust)

    def _advance_simulation(self, u, fext):
        """Replaces base_aviary.py:266-322: ``substeps`` fixed steps of dt/substeps, ZOH input.  With ``aero`` the drag
        force (held from the start of the control step) joins the external force and the ground effect rescales the
        thrusts at the start of every sub-step (base_aviary.py:293-305)."""
        f = rm.F_BY_SYSTEM[self.system]
        step = rm.INTEGRATORS[self.integrator]
        h = self.CTRL_TIMESTEP / self.substeps
        x = list(self.state)
        if self.drag and self.system != 'cartpole':
            d = rm.drag_force(self.system, x, u, self.params)
            fext = tuple(fext[i] + d[i] for i in range(3))
        for _ in range(self.substeps):
            ue = rm.ground_effect_thrusts(self.system, x, u, self.params) if (self.ground_effect and self.system != 'cartpole') else u
            x = step(f, x, ue, self.params, h, fext)
        self.state = np.array(x)

    def _get_observation(self):
        """quadrotor.py:844-884 + benchmark_env.py:406-420."""
        obs = deepcopy(self.state)
        if self.obs_noise is not None:
            obs = obs + self.obs_noise
        next_step = self.ctrl_step_counter + 1
        if self.COST == 'rl_reward' and self.TASK == 'traj_tracking' and self.obs_goal_horizon > 0:
            idx = [min(next_step + i, self.X_GOAL.shape[0] - 1) for i in range(self.obs_goal_horizon)]
            obs = np.concatenate([obs, self.X_GOAL[idx].flatten()])
        elif self.COST == 'rl_reward' and self.TASK == 'stabilization' and self.obs_goal_horizon > 0:
            obs = np.concatenate([obs, self.X_GOAL.flatten()])
        return obs

    def _goal_now(self):
        if self.TASK == 'stabilization':
            return self.X_GOAL
        return self.X_GOAL[min(self.ctrl_step_counter, self.X_GOAL.shape[0] - 1)]

    def _get_info(self):
        """quadrotor.py:1011-1025."""
        err = (self.state - self._goal_now()) * self.info_mse_metric_state_weight
        return {'mse': float(np.sum(err ** 2))}

    def _get_done(self):
        """quadrotor.py:956-993."""
        if self.TASK == 'stabilization' and self.COST == 'quadratic':
            self.goal_reached = bool(np.linalg.norm(self.state - self.X_GOAL)
                                     < self.TASK_INFO['stabilization_goal_tolerance'])
            if self.goal_reached:
                return True
        if self.done_on_out_of_bound:
            if self.system == 'cartpole':
                x, th = self.state[0], self.state[2]
                if x < -self.x_threshold or x > self.x_threshold or \
                        th < -self.theta_threshold_radians or th > self.theta_threshold_radians:
                    return True
            else:
                mask = {1: [1, 0], 2: [1, 0, 1, 0, 1, 0], 3: [1, 0, 1, 0, 1, 0, 1, 1, 1, 0, 0, 0]}[self.system]
                oob = np.logical_or(self.state < self.state_space_low, self.state > self.state_space_high)
                if np.any(oob * np.array(mask)):
                    return True
        return False

    def _get_reward(self):
        """quadrotor.py:886-930."""
        act = np.asarray(self.current_preprocessed_action)
        if self.COST == 'rl_reward':
            act_err = act - self.U_GOAL
            state_err = self.state - self._goal_now()           # :900-908 (clamped index for tracking)
            dist = np.sum(self.rew_state_weight * state_err * state_err)
            dist += np.sum(self.rew_act_weight * act_err * act_err)
            rew = -dist
            if self.rew_exponential:
                rew = np.exp(rew)
            return float(rew)
        if self.COST == 'quadratic':
            if self.TASK == 'stabilization':
                xr = self.X_GOAL
            else:
                xr = self.X_GOAL[self.ctrl_step_counter, :]     # :926 (unclamped)
            ex = self.state - xr
            eu = act - self.U_GOAL
            return float(-1 * (0.5 * ex @ self.Q @ ex + 0.5 * eu @ self.R @ eu))   # quadrotor.py:682
        raise NotImplementedError('cost not on the hot path: ' + self.COST)

    def after_step(self, obs, rew, done, info):
        """benchmark_env.py:422-463."""
        self.pyb_step_counter += self.PYB_STEPS_PER_CTRL
        self.ctrl_step_counter += 1
        c_value = None
        if self.constraints is not None:
            c_value = self.constraints.get_values(self)
            info['constraint_values'] = c_value
            if self.constraints.is_violated(self, c_value=c_value):
                info['constraint_violation'] = 1
                if self.DONE_ON_VIOLATION:
                    done = True
            else:
                info['constraint_violation'] = 0
        if self.COST == 'rl_reward':
            if self.constraints is not None and self.use_constraint_penalty and \
                    self.constraints.is_almost_active(self, c_value=c_value):
                rew += self.constraint_penalty
        if self.ctrl_step_counter >= self.CTRL_STEPS:
            info['TimeLimit.truncated'] = not done
            done = True
        return obs, rew, done, info
