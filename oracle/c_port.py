"""ctypes wrapper of the plain-C oracle (``oracle/c/libscg_oracle.so``).

Test infrastructure -- see ``oracle/__init__.py``.  ``CPort.from_ref_env(ref)`` copies the constants
of a (Python, float64) ``RefEnv`` into the C config so the two can be compared on identical inputs
(tests/test_oracle_c_port.py) and so ``bench.py`` can time the same workload on the host cores.
Supports what the benchmark configs use: RK4/Euler, rl_reward / diagonal quadratic cost, bounded
(``+-I``) constraint rows, out-of-bound / goal / time-limit termination, auto-reset to given states.
"""
import ctypes as C
import os

import numpy as np

_dbl, _i32 = C.c_double, C.c_int32


class ScgoConfig(C.Structure):
    _fields_ = [
        ('system', _i32), ('euler', _i32), ('substeps', _i32), ('tracking', _i32), ('quadratic', _i32),
        ('rew_exponential', _i32), ('ctrl_steps', _i32), ('motor_map', _i32), ('done_on_oob', _i32),
        ('done_on_violation', _i32), ('n_bounds', _i32), ('ref_rows', _i32),
        ('dt', _dbl),
        ('m', _dbl), ('g', _dbl), ('l', _dbl), ('ixx', _dbl), ('iyy', _dbl), ('izz', _dbl), ('kf', _dbl),
        ('km', _dbl), ('pwm_scale', _dbl), ('pwm_const', _dbl), ('pwm_min', _dbl), ('pwm_max', _dbl),
        ('cp_l', _dbl), ('cp_m', _dbl), ('cp_M', _dbl),
        ('act_low', _dbl * 4), ('act_high', _dbl * 4), ('u_goal', _dbl * 4),
        ('x_goal', _dbl * 12), ('w_state', _dbl * 12), ('w_act', _dbl * 4),
        ('oob_low', _dbl * 12), ('oob_high', _dbl * 12),
        ('goal_tol', _dbl),
        ('bound_dim', _i32 * 32), ('bound_sign', _dbl * 32), ('bound_b', _dbl * 32),
    ]


def library_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), 'c', 'libscg_oracle.so')


_LIB = None


def load():
    global _LIB
    if _LIB is None:
        lib = C.CDLL(library_path())
        lib.scgo_step.argtypes = [C.POINTER(ScgoConfig), C.c_int64] + [C.c_void_p] * 9 + [_i32]
        lib.scgo_step.restype = None
        lib.scgo_max_threads.restype = C.c_int
        _LIB = lib
    return _LIB


class CPort:
    def __init__(self, cfg, nx, nu, ref_table):
        self.lib = load()
        self.cfg, self.nx, self.nu = cfg, nx, nu
        self.ref = np.ascontiguousarray(ref_table, dtype=np.float64) if ref_table is not None else np.zeros((1, nx))

    @classmethod
    def from_ref_env(cls, r):
        c = ScgoConfig()
        c.system = 4 if r.system == 'cartpole' else int(r.system)
        c.euler = int(r.integrator == 'euler')
        c.substeps = r.substeps
        c.tracking = int(r.TASK == 'traj_tracking')
        c.quadratic = int(r.COST == 'quadratic')
        c.rew_exponential = int(bool(r.rew_exponential))
        c.ctrl_steps = int(r.CTRL_STEPS)
        c.motor_map = int(bool(r.use_motor_map))
        c.done_on_oob = int(bool(r.done_on_out_of_bound))
        c.done_on_violation = int(bool(r.DONE_ON_VIOLATION))
        c.dt = r.CTRL_TIMESTEP
        p = r.params
        if r.system == 'cartpole':
            c.cp_l, c.cp_m, c.cp_M, c.g = p.l, p.m, p.M, p.g
            c.m = c.ixx = c.iyy = c.izz = c.kf = c.km = c.pwm_scale = 1.0
            lo = [-r.x_threshold, -np.inf, -r.theta_threshold_radians, -np.inf]
            hi = [-v for v in lo]
        else:
            c.m, c.g, c.l, c.ixx, c.iyy, c.izz, c.kf, c.km = p.m, p.g, p.l, p.ixx, p.iyy, p.izz, p.kf, p.km
            c.pwm_scale, c.pwm_const, c.pwm_min, c.pwm_max = p.pwm2rpm_scale, p.pwm2rpm_const, p.min_pwm, p.max_pwm
            c.cp_l = c.cp_m = c.cp_M = 1.0
            mask = {1: [1, 0], 2: [1, 0, 1, 0, 1, 0], 3: [1, 0, 1, 0, 1, 0, 1, 1, 1, 0, 0, 0]}[r.system]
            lo = [r.state_space_low[j] if mask[j] else -np.inf for j in range(r.nx)]
            hi = [r.state_space_high[j] if mask[j] else np.inf for j in range(r.nx)]
        for j in range(r.nu):
            c.act_low[j], c.act_high[j], c.u_goal[j] = r.action_space_low[j], r.action_space_high[j], r.U_GOAL[j]
        if r.COST == 'quadratic':
            assert np.allclose(r.Q, np.diag(np.diag(r.Q))) and np.allclose(r.R, np.diag(np.diag(r.R)))
            ws, wa = np.diag(r.Q), np.diag(r.R)
        else:
            ws = np.broadcast_to(r.rew_state_weight, (r.nx,))
            wa = np.broadcast_to(r.rew_act_weight, (r.nu,))
        for j in range(r.nx):
            c.w_state[j], c.oob_low[j], c.oob_high[j] = ws[j], lo[j], hi[j]
            if r.TASK == 'stabilization':
                c.x_goal[j] = r.X_GOAL[j]
        for j in range(r.nu):
            c.w_act[j] = wa[j]
        c.goal_tol = r.TASK_INFO['stabilization_goal_tolerance']
        nb = 0
        if r.constraints is not None:
            for con in r.constraints.constraints:
                off = 0 if con.constrained_variable == 'state' else r.nx
                AF = con.A @ con.constraint_filter
                for k in range(AF.shape[0]):
                    nz = np.nonzero(AF[k])[0]
                    assert len(nz) == 1 and abs(AF[k, nz[0]]) == 1.0, 'C port handles bounded rows only'
                    c.bound_dim[nb], c.bound_sign[nb], c.bound_b[nb] = off + nz[0], AF[k, nz[0]], con.b[k]
                    nb += 1
        c.n_bounds = nb
        ref = r.X_GOAL if r.TASK == 'traj_tracking' else None
        c.ref_rows = ref.shape[0] if ref is not None else 1
        return cls(c, r.nx, r.nu, ref)

    def step(self, action, state, t, reset_state, want_constraints=False, num_threads=0):
        """In-place on ``state`` (n, nx) float64 and ``t`` (n,) int32.  Returns (reward, done, c_val, viol)."""
        n = state.shape[0]
        assert state.flags.c_contiguous and state.dtype == np.float64 and t.dtype == np.int32
        action = np.ascontiguousarray(action, dtype=np.float64)
        reset_state = np.ascontiguousarray(reset_state, dtype=np.float64)
        rew = np.empty(n)
        done = np.empty(n, dtype=np.uint8)
        cv = np.empty((n, max(self.cfg.n_bounds, 1))) if want_constraints else None
        viol = np.empty(n, dtype=np.uint8) if want_constraints else None
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        self.lib.scgo_step(C.byref(self.cfg), n, vp(action), vp(state), vp(t), vp(rew), vp(done), vp(self.ref),
                           vp(reset_state), vp(cv), vp(viol), num_threads)
        return rew, done, cv, viol

    def max_threads(self):
        return self.lib.scgo_max_threads()
