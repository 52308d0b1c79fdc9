"""The C4 step (Quadrotor-3D, traj_tracking, quadratic cost, RK4, auto-reset to given states) vectorised over the env axis in
NumPy float64 -- SURVEY.md 8(d) baseline (ii), "CPU-batched": the same operations as ref_env.RefEnv.step in the same order,
one array operation per scalar operation.  Test infrastructure (see oracle/__init__.py): checked against the C port / the
scalar oracle in tests/test_oracle_pinning.py, timed by bench.py's cpu_baseline.  One configuration only, on purpose.
"""
import numpy as np


class BatchedC4:
    def __init__(self, ref):
        """``ref``: a RefEnv(system=3, task='traj_tracking', cost='quadratic') whose constants are taken over."""
        assert ref.system == 3 and ref.TASK == 'traj_tracking' and ref.COST == 'quadratic' and ref.integrator == 'rk4' and ref.substeps == 1
        p = ref.params
        self.p, self.dt = p, ref.CTRL_TIMESTEP
        self.lo, self.hi = np.asarray(ref.action_space_low, dtype=np.float64), np.asarray(ref.action_space_high, dtype=np.float64)
        self.x_goal, self.u_goal = np.asarray(ref.X_GOAL, dtype=np.float64), np.asarray(ref.U_GOAL, dtype=np.float64)
        self.Q, self.R = np.asarray(ref.Q, dtype=np.float64), np.asarray(ref.R, dtype=np.float64)
        self.s_lo, self.s_hi = np.asarray(ref.state_space_low, dtype=np.float64), np.asarray(ref.state_space_high, dtype=np.float64)
        self.mask = np.array([1, 0, 1, 0, 1, 0, 1, 1, 1, 0, 0, 0], dtype=bool)            # quadrotor.py:979-993
        self.ctrl_steps = int(ref.CTRL_STEPS)
        self.motor_map, self.done_oob = ref.use_motor_map, ref.done_on_out_of_bound

    def f(self, x, u):
        """quadrotor.py:624-674 on [N, 12] / [N, 4] (ref_model.f_quad3d)."""
        p = self.p
        phi, th, psi = x[:, 6], x[:, 7], x[:, 8]
        pp, q, r = x[:, 9], x[:, 10], x[:, 11]
        f1, f2, f3, f4 = u[:, 0], u[:, 1], u[:, 2], u[:, 3]
        cz, sz, cy, sy, cx, sx = np.cos(psi), np.sin(psi), np.cos(th), np.sin(th), np.cos(phi), np.sin(phi)
        thrust = f1 + f2 + f3 + f4
        r02, r12, r22 = cz * sy * cx + sz * sx, sz * sy * cx - cz * sx, cy * cx          # third column of Rz Ry Rx
        s2 = np.sqrt(2.0)
        gamma = p.km / p.kf
        m0, m1, m2 = p.l / s2 * (f1 + f2 - f3 - f4), p.l / s2 * (-f1 + f2 + f3 - f4), gamma * (f1 - f2 + f3 - f4)
        j0, j1, j2 = p.ixx * pp, p.iyy * q, p.izz * r
        tth = np.tan(th)
        out = np.empty_like(x)
        out[:, 0], out[:, 2], out[:, 4] = x[:, 1], x[:, 3], x[:, 5]
        out[:, 1], out[:, 3], out[:, 5] = r02 * thrust / p.m, r12 * thrust / p.m, r22 * thrust / p.m - p.g
        out[:, 6] = pp + sx * tth * q + cx * tth * r
        out[:, 7] = cx * q - sx * r
        out[:, 8] = sx / cy * q + cx / cy * r
        out[:, 9] = (m0 - (q * j2 - r * j1)) / p.ixx
        out[:, 10] = (m1 - (r * j0 - pp * j2)) / p.iyy
        out[:, 11] = (m2 - (pp * j1 - q * j0)) / p.izz
        return out

    def step(self, action, state, t, reset_state):
        """In place on ``state`` [N, 12] float64 and ``t`` [N] int32; -> (reward, done).  Finished envs restart from
        ``reset_state`` (dummy_vec_env.py:41-47)."""
        p = self.p
        thrust = np.clip(action, self.lo, self.hi)                                       # quadrotor.py:830
        u = thrust
        if self.motor_map:                                                               # quadrotor_utils.py:21-67, base_aviary.py:420
            pwm = (np.sqrt(np.maximum(thrust, 0.0) / p.kf) - p.pwm2rpm_const) / p.pwm2rpm_scale
            rpm = p.pwm2rpm_scale * np.clip(pwm, p.min_pwm, p.max_pwm) + p.pwm2rpm_const
            u = p.kf * rpm * rpm
        dt = self.dt
        k1 = self.f(state, u)                                                            # mpc_utils.py:79-101
        k2 = self.f(state + dt / 2 * k1, u)
        k3 = self.f(state + dt / 2 * k2, u)
        k4 = self.f(state + dt * k3, u)
        x = state + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        done = np.zeros(x.shape[0], dtype=bool)
        if self.done_oob:
            with np.errstate(invalid='ignore'):
                oob = ((x < self.s_lo) | (x > self.s_hi) | ~np.isfinite(x)) & self.mask
            done = oob.any(axis=1)
        ex = x - self.x_goal[np.minimum(t, self.x_goal.shape[0] - 1)]                    # quadrotor.py:926
        eu = thrust - self.u_goal
        rew = -(0.5 * np.einsum('ni,ij,nj->n', ex, self.Q, ex) + 0.5 * np.einsum('ni,ij,nj->n', eu, self.R, eu))
        t1 = t + 1
        done |= t1 >= self.ctrl_steps                                                    # benchmark_env.py:457-460
        state[...] = np.where(done[:, None], reset_state, x)
        t[...] = np.where(done, 0, t1)
        return rew, done
