"""Batched 1-D / 2-D / 3-D quadrotor -- host-side mirror of the reference's ``Quadrotor`` env.

Constructor keywords, attribute names and yaml keys follow
/root/reference/safe_control_gym/envs/gym_pybullet_drones/quadrotor.py (``__init__`` :148-329, spaces
:699-806, default ``quadrotor.yaml``) and ``base_aviary.py`` (:66-147 constants, :612-651 URDF
values).  The per-step work -- ``_preprocess_control`` (:815-842), the dynamics (:594-676) under
RK4 (controllers/mpc/mpc_utils.py:79-101), ``_get_observation`` (:844-884), ``_get_reward``
(:886-930), ``_get_done`` (:956-993), the ``mse`` of ``_get_info`` (:1011-1025) and ``after_step``
(benchmark_env.py:422-463) -- runs in the fused HIP kernel; this file only prepares its constants.

Out of scope here (needs a contact/ray-casting physics engine): gates, obstacles, collisions, the
competition cost (:370-437, :933-954, :1046-1132).
"""
import math
from copy import deepcopy
from enum import IntEnum

import weakref

import numpy as np

from safe_control_gym_amd.envs.benchmark_env import BenchmarkEnv, Cost, Task
from safe_control_gym_amd.envs.constraints import GENERAL_CONSTRAINTS
from safe_control_gym_amd.envs.spaces import Box
from safe_control_gym_amd.math_and_models.symbolic_systems import AnalyticModel
from safe_control_gym_amd.math_and_models.transformations import transform_trajectory


class QuadType(IntEnum):
    """quadrotor_utils.py:11-18."""
    ONE_D = 1
    TWO_D = 2
    THREE_D = 3


# assets/cf2x.urdf:5,12,13 (parsed by base_aviary.py:612-651 in the reference)
CF2X = dict(MASS=0.03454, L=0.0397, THRUST2WEIGHT_RATIO=2.25, IXX=1.4e-5, IYY=1.4e-5, IZZ=2.17e-5,
            KF=3.16e-10, KM=7.94e-12, PWM2RPM_SCALE=0.2685, PWM2RPM_CONST=4070.3, MIN_PWM=20000.0,
            MAX_PWM=65535.0, GND_EFF_COEFF=11.36859, PROP_RADIUS=2.31348e-2, DRAG_COEFF_XY=9.1785e-7,
            DRAG_COEFF_Z=10.311e-7)            # assets/cf2x.urdf:5

_F32MAX = float(np.finfo(np.float32).max)


def _uniform(lo, hi):
    return {'distrib': 'uniform', 'low': lo, 'high': hi}


class Quadrotor(BenchmarkEnv):
    NAME = 'quadrotor'
    AVAILABLE_CONSTRAINTS = deepcopy(GENERAL_CONSTRAINTS)

    INIT_STATE_RAND_INFO = {                    # quadrotor.py:68-134
        'init_x': _uniform(-0.5, 0.5), 'init_x_dot': _uniform(-0.01, 0.01),
        'init_y': _uniform(-0.5, 0.5), 'init_y_dot': _uniform(-0.01, 0.01),
        'init_z': _uniform(0.1, 1.5), 'init_z_dot': _uniform(-0.01, 0.01),
        'init_phi': _uniform(-0.3, 0.3), 'init_theta': _uniform(-0.3, 0.3), 'init_psi': _uniform(-0.3, 0.3),
        'init_p': _uniform(-0.01, 0.01), 'init_theta_dot': _uniform(-0.01, 0.01),
        'init_q': _uniform(-0.01, 0.01), 'init_r': _uniform(-0.01, 0.01),
    }
    INERTIAL_PROP_RAND_INFO = {                 # quadrotor.py:45-66
        'M': _uniform(0.022, 0.032), 'Ixx': _uniform(1.3e-5, 1.5e-5), 'Iyy': _uniform(1.3e-5, 1.5e-5),
        'Izz': _uniform(2.07e-5, 2.27e-5),
    }
    TASK_INFO = {                               # quadrotor.py:136-146
        'stabilization_goal': [0, 1], 'stabilization_goal_tolerance': 0.05, 'trajectory_type': 'circle',
        'num_cycles': 1, 'trajectory_plane': 'zx', 'trajectory_position_offset': [0.5, 0],
        'trajectory_scale': -0.5, 'proj_point': [0, 0, 0.5], 'proj_normal': [0, 1, 1],
    }
    _INERTIAL_KEYS = ('M', 'Ixx', 'Iyy', 'Izz')
    _LABELS = {                                 # quadrotor.py:207-212
        QuadType.ONE_D: ['init_x', 'init_x_dot'],
        QuadType.TWO_D: ['init_x', 'init_x_dot', 'init_z', 'init_z_dot', 'init_theta', 'init_theta_dot'],
        QuadType.THREE_D: ['init_x', 'init_x_dot', 'init_y', 'init_y_dot', 'init_z', 'init_z_dot',
                           'init_phi', 'init_theta', 'init_psi', 'init_p', 'init_q', 'init_r'],
    }

    def __init__(self, init_state=None, inertial_prop=None, quad_type=QuadType.TWO_D, norm_act_scale=0.1,
                 obs_goal_horizon=0, rew_state_weight=1.0, rew_act_weight=0.0001, rew_exponential=True,
                 done_on_out_of_bound=True, info_mse_metric_state_weight=None, **kwargs):
        self.QUAD_TYPE = QuadType(quad_type)
        self.SYSTEM = int(self.QUAD_TYPE)
        # gates / obstacles (quadrotor.py:370-437,1046-1132) need a contact / ray-casting physics engine: the analytic engine flies
        # through them.  The keys are accepted (the reference's configs carry them) and the caller is told.
        if kwargs.get('gates') or kwargs.get('obstacles') or kwargs.get('randomized_gates_and_obstacles'):
            import warnings
            warnings.warn('[WARNING] gates / obstacles (and done_on_collision / done_on_completion) need a contact physics engine: '
                          'ignored by the analytic engine (DESIGN.md section 1).', stacklevel=3)
        self.DISTURBANCE_MODES = {'observation': {'dim': -1}, 'action': {'dim': -1}, 'dynamics': {'dim': -1}}
        self.norm_act_scale = norm_act_scale
        self.obs_goal_horizon = obs_goal_horizon
        self.rew_state_weight = np.array(rew_state_weight, ndmin=1, dtype=float)
        self.rew_act_weight = np.array(rew_act_weight, ndmin=1, dtype=float)
        self.rew_exponential = rew_exponential
        self.done_on_out_of_bound = done_on_out_of_bound
        nx = {1: 2, 2: 6, 3: 12}[self.SYSTEM]
        if info_mse_metric_state_weight is None:        # quadrotor.py:185-193
            w = {1: [1, 0], 2: [1, 0, 1, 0, 0, 0], 3: [1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0]}[self.SYSTEM]
        elif len(info_mse_metric_state_weight) == nx:
            w = info_mse_metric_state_weight
        else:
            raise ValueError('[ERROR] in Quadrotor.__init__(), wrong info_mse_metric_state_weight argument size.')
        self.info_mse_metric_state_weight = np.array(w, ndmin=1, dtype=float)
        for arr, name, size in ((self.rew_state_weight, 'rew_state_weight', nx),
                                (self.rew_act_weight, 'rew_act_weight', {1: 1, 2: 2, 3: 4}[self.SYSTEM])):
            if arr.shape[0] not in (1, size):
                raise ValueError('[ERROR] in Quadrotor.__init__(), wrong %s size.' % name)
        # base_aviary.py:90-147 constants
        self.GRAVITY_ACC = 9.8
        self.MASS, self.L = CF2X['MASS'], CF2X['L']
        self.J = np.diag([CF2X['IXX'], CF2X['IYY'], CF2X['IZZ']])
        self.KF, self.KM = CF2X['KF'], CF2X['KM']
        self.PWM2RPM_SCALE, self.PWM2RPM_CONST = CF2X['PWM2RPM_SCALE'], CF2X['PWM2RPM_CONST']
        self.MIN_PWM, self.MAX_PWM = CF2X['MIN_PWM'], CF2X['MAX_PWM']
        self.GROUND_PLANE_Z = 0
        # inertial overrides (quadrotor.py:242-256) -- before the spaces, which use MASS
        if inertial_prop is None:
            pass
        elif isinstance(inertial_prop, dict):
            self.MASS = inertial_prop.get('M', self.MASS)
            self.J[0, 0] = inertial_prop.get('Ixx', self.J[0, 0])
            self.J[1, 1] = inertial_prop.get('Iyy', self.J[1, 1])
            self.J[2, 2] = inertial_prop.get('Izz', self.J[2, 2])
        elif self.QUAD_TYPE == QuadType.ONE_D and np.array(inertial_prop).shape == (1,):
            self.MASS = inertial_prop[0]
        elif self.QUAD_TYPE == QuadType.TWO_D and np.array(inertial_prop).shape == (2,):
            self.MASS, self.J[1, 1] = inertial_prop
        elif self.QUAD_TYPE == QuadType.THREE_D and np.array(inertial_prop).shape == (4,):
            self.MASS, self.J[0, 0], self.J[1, 1], self.J[2, 2] = inertial_prop
        else:
            raise ValueError('[ERROR] in Quadrotor.__init__(), inertial_prop incorrect format.')
        self.GRAVITY = self.GRAVITY_ACC * self.MASS
        self.HOVER_RPM = np.sqrt(self.GRAVITY / (4 * self.KF))
        super().__init__(init_state=init_state, inertial_prop=inertial_prop, **kwargs)

        labels = self._LABELS[self.QUAD_TYPE]
        self.INIT_STATE_LABELS = self._LABELS
        if init_state is None:                          # quadrotor.py:213-224
            nominal = np.zeros(nx)
        elif isinstance(init_state, np.ndarray):
            nominal = np.array(init_state, dtype=float).reshape(nx)
        elif isinstance(init_state, dict):
            nominal = np.array([init_state.get(k, 0.) for k in labels], dtype=float)
        else:
            raise ValueError('[ERROR] in Quadrotor.__init__(), init_state incorrect format.')
        self._init_nominal = nominal
        for k, name in enumerate(labels):
            setattr(self, name.upper(), nominal[k])
        self.INIT_STATE_RAND_INFO = {k: v for k, v in self.INIT_STATE_RAND_INFO.items() if k in labels}  # :227-229
        # inertial randomisation only of what the quad type has (quadrotor.py:231-239)
        keep = {QuadType.ONE_D: ('M',), QuadType.TWO_D: ('M', 'Iyy'), QuadType.THREE_D: ('M', 'Ixx', 'Iyy', 'Izz')}
        self.INERTIAL_PROP_RAND_INFO = {k: v for k, v in self.INERTIAL_PROP_RAND_INFO.items()
                                        if k in keep[self.QUAD_TYPE]}
        self._setup_symbolic()
        # goals (quadrotor.py:261-329)
        self.U_GOAL = np.ones(self.action_dim) * self.MASS * self.GRAVITY_ACC / self.action_dim
        ti = self.TASK_INFO
        if self.TASK == Task.STABILIZATION:
            g = ti['stabilization_goal']
            if self.QUAD_TYPE == QuadType.ONE_D:
                self.X_GOAL = np.hstack([g[1], 0.0])
            elif self.QUAD_TYPE == QuadType.TWO_D:
                self.X_GOAL = np.hstack([g[0], 0.0, g[1], 0.0, 0.0, 0.0])
            else:
                if len(g) < 3:      # the reference indexes goal[2] (quadrotor.py:277) -> the same IndexError, explained
                    raise IndexError("[ERROR] 3-D stabilization needs task_info['stabilization_goal'] = [x, y, z] "
                                     '(the default TASK_INFO holds the 2-D goal [0, 1]).')
                self.X_GOAL = np.hstack([g[0], 0.0, g[1], 0.0, g[2], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
        else:
            pos, vel, _ = self._generate_trajectory(traj_type=ti['trajectory_type'],
                                                    traj_length=self.EPISODE_LEN_SEC,
                                                    num_cycles=ti['num_cycles'],
                                                    traj_plane=ti['trajectory_plane'],
                                                    position_offset=ti['trajectory_position_offset'],
                                                    scaling=ti['trajectory_scale'],
                                                    sample_time=self.CTRL_TIMESTEP)
            z = np.zeros(pos.shape[0])
            if self.QUAD_TYPE == QuadType.ONE_D:
                cols = [pos[:, 2], vel[:, 2]]
            elif self.QUAD_TYPE == QuadType.TWO_D:
                cols = [pos[:, 0], vel[:, 0], pos[:, 2], vel[:, 2], z, z]
            else:
                pos, vel = transform_trajectory(pos, vel, {'point': ti['proj_point'], 'normal': ti['proj_normal']})
                cols = [pos[:, 0], vel[:, 0], pos[:, 1], vel[:, 1], pos[:, 2], vel[:, 2], z, z, z, z, z, z]
            self.X_GOAL = np.stack(cols, axis=1)
        self.X_EQ = np.zeros(self.state_dim)
        self.U_EQ = self.U_GOAL

    # ------------------------------------------------------------------ spaces
    def _set_action_space(self):
        """quadrotor.py:699-733."""
        nu = {1: 1, 2: 2, 3: 4}[self.SYSTEM]
        self.ACTION_LABELS = [['T'], ['T1', 'T2'], ['T1', 'T2', 'T3', 'T4']][self.SYSTEM - 1]
        self.ACTION_UNITS = ['-' if self.NORMALIZED_RL_ACTION_SPACE else 'N'] * nu
        if self.NORMALIZED_RL_ACTION_SPACE:
            self.hover_thrust = self.GRAVITY_ACC * self.MASS / nu
            self.action_space = Box(low=-np.ones(nu), high=np.ones(nu), dtype=np.float32)
        else:
            n_motors = 4 / nu
            a_low = self.KF * n_motors * (self.PWM2RPM_SCALE * self.MIN_PWM + self.PWM2RPM_CONST) ** 2
            a_high = self.KF * n_motors * (self.PWM2RPM_SCALE * self.MAX_PWM + self.PWM2RPM_CONST) ** 2
            self.action_space = Box(low=np.full(nu, a_low, np.float32), high=np.full(nu, a_high, np.float32),
                                    dtype=np.float32)

    def _set_observation_space(self):
        """quadrotor.py:735-806."""
        self.x_threshold = self.y_threshold = 5
        self.z_threshold = 2.5
        self.phi_threshold_radians = self.theta_threshold_radians = 85 * math.pi / 180
        self.psi_threshold_radians = 180 * math.pi / 180
        X, Z, A, P, M, G = (self.x_threshold, self.z_threshold, self.theta_threshold_radians,
                            self.psi_threshold_radians, _F32MAX, self.GROUND_PLANE_Z)
        if self.QUAD_TYPE == QuadType.ONE_D:
            low, high = [G, -M], [Z, M]
            self.STATE_LABELS, self.STATE_UNITS = ['z', 'z_dot'], ['m', 'm/s']
        elif self.QUAD_TYPE == QuadType.TWO_D:
            low, high = [-X, -M, G, -M, -A, -M], [X, M, Z, M, A, M]
            self.STATE_LABELS = ['x', 'x_dot', 'z', 'z_dot', 'theta', 'theta_dot']
            self.STATE_UNITS = ['m', 'm/s', 'm', 'm/s', 'rad', 'rad/s']
        else:
            low = [-X, -M, -X, -M, G, -M, -A, -A, -P, -M, -M, -M]
            high = [X, M, X, M, Z, M, A, A, P, M, M, M]
            self.STATE_LABELS = ['x', 'x_dot', 'y', 'y_dot', 'z', 'z_dot', 'phi', 'theta', 'psi', 'p', 'q', 'r']
            self.STATE_UNITS = ['m', 'm/s', 'm', 'm/s', 'm', 'm/s', 'rad', 'rad', 'rad', 'rad/s', 'rad/s', 'rad/s']
        low, high = np.array(low, dtype=np.float64), np.array(high, dtype=np.float64)
        self.state_space = Box(low=low, high=high, dtype=np.float32)
        mul = 1
        if self.COST == Cost.RL_REWARD and self.obs_goal_horizon > 0:
            mul = 1 + self.obs_goal_horizon if self.TASK == Task.TRAJ_TRACKING else 2
        self.observation_space = Box(low=np.concatenate([low] * mul), high=np.concatenate([high] * mul),
                                     dtype=np.float32)

    def _setup_disturbances(self):
        """quadrotor.py:808-813."""
        self.DISTURBANCE_MODES['observation']['dim'] = self.state_dim
        self.DISTURBANCE_MODES['action']['dim'] = self.action_dim
        self.DISTURBANCE_MODES['dynamics']['dim'] = int(self.QUAD_TYPE)
        super()._setup_disturbances()

    def _setup_symbolic(self):
        """quadrotor.py:581-697 -> closed-form prior model (no CasADi)."""
        consts = dict(m=self.MASS, g=self.GRAVITY_ACC, l=self.L, Ixx=self.J[0, 0], Iyy=self.J[1, 1],
                      Izz=self.J[2, 2], gamma=self.KM / self.KF)
        self.symbolic = AnalyticModel({1: 'quad1d', 2: 'quad2d', 3: 'quad3d'}[self.SYSTEM], consts,
                                      self.CTRL_TIMESTEP)
        self.symbolic.device_backend = weakref.proxy(self)

    def _get_reset_info(self):
        """quadrotor.py:1134-1200 (the keys that exist without gates / obstacles / PyBullet)."""
        info = super()._get_reset_info()
        info['nominal_physical_parameters'] = {'quadrotor_mass': self.MASS, 'quadrotor_ixx_inertia': self.J[0, 0],
                                               'quadrotor_iyy_inertia': self.J[1, 1], 'quadrotor_izz_inertia': self.J[2, 2]}
        info['quadrotor_kf'] = self.KF
        info['quadrotor_km'] = self.KM
        return info

    @property
    def OVERRIDDEN_QUAD_MASS(self):
        """quadrotor.py:466: the mass in force since the last reset -- per env ([N] device tensor) when the inertial properties
        are randomised, the nominal value otherwise."""
        if self.RANDOMIZED_INERTIAL_PROP and self._engine is not None and 'inertial' in self._bufs:
            return self.inertial_properties[:, 0]
        return self.MASS

    @property
    def OVERRIDDEN_QUAD_INERTIA(self):
        """quadrotor.py:467: (Ixx, Iyy, Izz) -- [N, 3] per env when randomised, the nominal list otherwise."""
        if self.RANDOMIZED_INERTIAL_PROP and self._engine is not None and 'inertial' in self._bufs:
            return self.inertial_properties[:, 1:4]
        return [self.J[0, 0], self.J[1, 1], self.J[2, 2]]

    def _inertial_nominal(self):
        return [self.MASS, self.J[0, 0], self.J[1, 1], self.J[2, 2]]

    # ------------------------------------------------------------------ C-ABI constants
    def _fill_system_config(self, cfg):
        cfg.mass, cfg.gravity, cfg.arm = self.MASS, self.GRAVITY_ACC, self.L
        cfg.ixx, cfg.iyy, cfg.izz = self.J[0, 0], self.J[1, 1], self.J[2, 2]
        cfg.kf, cfg.km = self.KF, self.KM
        cfg.pwm2rpm_scale, cfg.pwm2rpm_const = self.PWM2RPM_SCALE, self.PWM2RPM_CONST
        cfg.pwm_min, cfg.pwm_max = self.MIN_PWM, self.MAX_PWM
        # Physics.PYB_GND / PYB_DRAG / PYB_GND_DRAG_DW (base_aviary.py:35-45): aero terms of the analytic model; the
        # downwash of PYB_DW acts between drones of one env and does not exist for a single drone
        cfg.ground_effect = int(self.PHYSICS in ('pyb_gnd', 'pyb_gnd_drag_dw'))
        cfg.drag = int(self.PHYSICS in ('pyb_drag', 'pyb_gnd_drag_dw'))
        cfg.gnd_eff_coeff, cfg.prop_radius = CF2X['GND_EFF_COEFF'], CF2X['PROP_RADIUS']
        max_rpm2 = CF2X['THRUST2WEIGHT_RATIO'] * self.GRAVITY_ACC * CF2X['MASS'] / (4 * self.KF)      # base_aviary.py:141-147 (URDF mass)
        cfg.gnd_eff_h_clip = 0.25 * CF2X['PROP_RADIUS'] * math.sqrt(15 * max_rpm2 * self.KF * CF2X['GND_EFF_COEFF']
                                                                    / (4 * self.KF * max_rpm2))
        cfg.drag_coeff_xy, cfg.drag_coeff_z = CF2X['DRAG_COEFF_XY'], CF2X['DRAG_COEFF_Z']
        cfg.pole_length = cfg.pole_mass = cfg.cart_mass = 1.0
        cfg.norm_act_scale = self.norm_act_scale
        cfg.hover_thrust = getattr(self, 'hover_thrust', self.GRAVITY_ACC * self.MASS / self.action_dim)
        cfg.action_scale = 1.0
        mask = {1: [1, 0], 2: [1, 0, 1, 0, 1, 0], 3: [1, 0, 1, 0, 1, 0, 1, 1, 1, 0, 0, 0]}[self.SYSTEM]  # :981-985
        for j in range(self.state_dim):
            cfg.oob_low[j] = float(self.state_space.low[j]) if mask[j] else -math.inf
            cfg.oob_high[j] = float(self.state_space.high[j]) if mask[j] else math.inf
            cfg.init_state[j] = float(self._init_nominal[j])
        for j, label in enumerate(self._LABELS[self.QUAD_TYPE]):
            info = self.INIT_STATE_RAND_INFO.get(label) if self.RANDOMIZED_INIT else None
            if info is not None:
                from safe_control_gym_amd.envs.randomization import parse_distribution
                kind, lo, hi, prm = parse_distribution(info, label)                # benchmark_env.py:297-307
                cfg.init_rand_kind[j] = kind
                cfg.init_rand_low[j], cfg.init_rand_high[j] = lo, hi
                for q in range(3):
                    cfg.init_rand_p[j][q] = prm[q]
