"""Pins the CPU oracle (CPU-only tests).

The reference's own tests hold no golden vectors or known-answer tests for this path (SURVEY.md
section 4) and the reference cannot be imported here (casadi / pybullet / gym missing), so the oracle is
pinned by (a) the analytic known-answer tests of SURVEY.md section 8(c), (b) an independent sympy /
DOP853 derivation, (c) published Philox vectors, and (d) frozen golden fixtures of its own output.
"""
import json
import math
import os

import numpy as np
import pytest

from oracle import ref_model as rm
from oracle import independent as ind
from oracle import philox
from oracle.ref_env import RefEnv
from oracle.ref_vec import RefVecEnv
from oracle.ref_trajectory import generate_trajectory, transform_trajectory, projection_matrix
from helpers import PhiloxSampler

HERE = os.path.dirname(os.path.abspath(__file__))
P = rm.QuadParams()
DT = 1 / 60


def zeros_sampler(i, e):
    return np.zeros(12)


# ---- (0) the three models are one rigid body: the 1-D and 2-D restatements (quadrotor.py:594-621) equal the 3-D one (:624-674, whose
#      rigid-body terms are pinned to the reference's executed BaseAviary._dynamics: tests/test_reference_pins.py) on their subspaces
def test_planar_and_vertical_models_are_restrictions_of_the_3d_model():
    rng = np.random.default_rng(3)
    for _ in range(50):
        # 1-D: level attitude, the thrust T spread over the four motors
        z, zd, T = rng.uniform(0, 2), rng.uniform(-2, 2), rng.uniform(0.1, 0.6)
        x3 = [0.0] * 12
        x3[4], x3[5] = z, zd
        full = rm.f_quad3d(x3, [T / 4] * 4, P)
        assert np.allclose(rm.f_quad1d([z, zd], [T], P), [full[4], full[5]], rtol=1e-14, atol=1e-14)
        assert np.allclose([full[k] for k in (0, 1, 2, 3, 6, 7, 8, 9, 10, 11)], 0.0, atol=1e-12)          # nothing leaves the z axis
        # 2-D: pitch only; T1 = f1 + f4, T2 = f2 + f3 (the pairs whose difference is the pitch moment of quadrotor.py:668)
        x2 = rng.uniform(-1, 1, size=6)
        T1, T2 = rng.uniform(0.05, 0.3, size=2)
        x3 = [0.0] * 12
        x3[0], x3[1], x3[4], x3[5], x3[7], x3[10] = x2
        full = rm.f_quad3d(x3, [T1 / 2, T2 / 2, T2 / 2, T1 / 2], P)
        assert np.allclose(rm.f_quad2d(list(x2), [T1, T2], P), [full[0], full[1], full[4], full[5], full[7], full[10]], rtol=1e-13, atol=1e-13)
        assert np.allclose([full[k] for k in (2, 3, 6, 8, 9, 11)], 0.0, atol=1e-12)                           # and nothing leaves the x-z plane


# ---- (1) hover fixed point: level attitude, u = m g / nu each -> x_dot = 0 (quadrotor.py:262,328-329)
@pytest.mark.parametrize('system', [1, 2, 3])
def test_kat_hover_fixed_point(system):
    nx, nu = rm.NX[system], rm.NU[system]
    x = [0.0] * nx
    x[{1: 0, 2: 2, 3: 4}[system]] = 1.0
    u = [P.m * P.g / nu] * nu
    f = rm.F_BY_SYSTEM[system](x, u, P)
    assert np.allclose(f, 0.0, atol=1e-15)
    assert np.allclose(rm.rk4_step(rm.F_BY_SYSTEM[system], x, u, P, DT), x, atol=1e-15)


# ---- (2) free fall / constant thrust in 1-D: z(t) quadratic, which RK4 reproduces to rounding
def test_kat_1d_constant_acceleration():
    T = rm.action_bounds(1)[0]
    a = T / P.m - P.g
    x = [1.0, 0.3]
    for _ in range(10):
        x = rm.rk4_step(rm.f_quad1d, x, [T], P, DT)
    t = 10 * DT
    assert abs(x[0] - (1.0 + 0.3 * t + 0.5 * a * t * t)) < 1e-14
    assert abs(x[1] - (0.3 + a * t)) < 1e-14


# ---- (3) 3-D pure yaw couple at level attitude: r_dot = 4 gamma delta / Izz, psi quadratic, rest unchanged
def test_kat_3d_yaw_couple():
    h, d = P.m * P.g / 4, 0.003
    u = [h + d, h - d, h + d, h - d]
    x = [0.0] * 12
    x[4] = 1.0
    for _ in range(6):
        x = rm.rk4_step(rm.f_quad3d, x, u, P, DT)
    t = 6 * DT
    rdot = 4 * (P.km / P.kf) * d / P.izz
    assert abs(x[11] - rdot * t) < 1e-12
    assert abs(x[8] - 0.5 * rdot * t * t) < 1e-12
    assert np.allclose([x[0], x[1], x[2], x[3], x[4] - 1.0, x[5], x[6], x[7], x[9], x[10]], 0.0, atol=1e-13)


# ---- (4) 2-D pure pitch couple: theta_ddot = l (T2 - T1) / (Iyy sqrt 2)
def test_kat_2d_pitch_couple():
    u = [0.10, 0.11]
    d = rm.f_quad2d([0, 0, 1, 0, 0, 0], u, P)
    assert abs(d[5] - P.l * (u[1] - u[0]) / P.iyy / math.sqrt(2)) < 1e-12
    assert d[1] == 0.0 and abs(d[3] - ((u[0] + u[1]) / P.m - P.g)) < 1e-15


# ---- (5) action-bound constants of SURVEY.md row A2
def test_kat_action_bounds():
    assert rm.action_bounds(4) == pytest.approx((0.02816168745, 0.14834144334), rel=3e-8)
    assert rm.action_bounds(2) == pytest.approx((0.0563234, 0.2966829), rel=1e-6)
    assert rm.action_bounds(1) == pytest.approx((0.1126467, 0.5933658), rel=1e-6)
    # stored float32-rounded, as gym.spaces.Box(dtype=float32) does
    lo, hi = rm.action_bounds(4)
    assert lo == float(np.float32(lo)) and hi == float(np.float32(hi))
    assert P.m * P.g / 4 == pytest.approx(0.084623, rel=1e-6)


# ---- (6) the competition BoundedConstraint (getting_started.yaml:128-156) at x=(0,.,0,.,1,...)
COMPETITION = [{'constraint_form': 'default_constraint', 'constrained_variable': 'input'},
               {'constraint_form': 'bounded_constraint', 'constrained_variable': 'state', 'active_dims': [0, 2, 4],
                'upper_bounds': [3, 3, 2], 'lower_bounds': [-3, -3, -0.1]}]


def _env3(**kw):
    kw.setdefault('task_info', {'stabilization_goal': [0, 0, 1], 'stabilization_goal_tolerance': 0.05})
    return RefEnv(system=3, ctrl_freq=60, pyb_freq=240, randomized_init=False, reset_sampler=zeros_sampler, **kw)


def test_kat_bounded_and_default_constraints():
    env = _env3(constraints=COMPETITION, init_state=np.array([0, 0, 0, 0, 1.0, 0, 0, 0, 0, 0, 0, 0]))
    env.reset()
    _, _, _, info = env.step(env.U_GOAL)
    c = info['constraint_values']
    assert c.shape == (14,)
    a_low, a_high = rm.action_bounds(4)
    # (7) DefaultConstraint(input) at u = U_GOAL: (a_low - u) x4, (u - a_high) x4 -- lower rows first
    assert np.allclose(c[:4], a_low - 0.084623, atol=1e-6) and np.allclose(c[4:8], 0.084623 - a_high, atol=1e-6)
    assert np.allclose(c[8:], [-3, -3, -1.1, -3, -3, -1], atol=1e-7)
    assert info['constraint_violation'] == 0


# ---- (8) RL reward at the goal with a = U_GOAL is exp(0) = 1; quadratic cost there is 0
def test_kat_rewards_at_goal():
    x0 = np.array([0, 0, 0, 0, 1.0, 0, 0, 0, 0, 0, 0, 0])
    env = _env3(cost='rl_reward', init_state=x0)
    env.reset()
    _, rew, _, _ = env.step(env.U_GOAL)
    assert abs(rew - 1.0) < 1e-12
    env = _env3(cost='quadratic', init_state=x0)
    env.reset()
    _, rew, done, info = env.step(env.U_GOAL)
    assert abs(rew) < 1e-20 and done and info['goal_reached']       # within 0.05 of the goal -> done (:964-967)


# ---- (9) time limit: step 300 of a 5 s / 60 Hz episode
def test_kat_time_limit():
    env = _env3(cost='rl_reward', init_state=np.array([0, 0, 0, 0, 1.0, 0, 0, 0, 0, 0, 0, 0]))
    env.reset()
    for k in range(300):
        _, _, done, info = env.step(env.U_GOAL)
        if k < 299:
            assert not done and 'TimeLimit.truncated' not in info
    assert done and info['TimeLimit.truncated'] is True


# ---- (10) out-of-bound termination at z < 0 and |theta| > 85 deg (quadrotor.py:742-747,979-993)
def test_kat_out_of_bound_done():
    env = _env3(cost='rl_reward', init_state=np.array([0, 0, 0, 0, 0.001, -1.0, 0, 0, 0, 0, 0, 0]))
    env.reset()
    assert env.step(env.U_GOAL)[2]
    env = _env3(cost='rl_reward', init_state=np.array([0, 0, 0, 0, 1.0, 0, 0, 1.4836, 0, 0, 0, 0]))
    env.reset()
    assert env.step(env.U_GOAL)[2]
    env = _env3(cost='rl_reward', init_state=np.array([0, 0, 0, 0, 1.0, 0, 0, 1.40, 0, 0, 0, 0]))
    env.reset()
    assert not env.step(env.U_GOAL)[2]
    env = _env3(cost='rl_reward', done_on_out_of_bound=False,
                init_state=np.array([0, 0, 0, 0, 0.001, -1.0, 0, 0, 0, 0, 0, 0]))
    env.reset()
    assert not env.step(env.U_GOAL)[2]


# ---- the Euler-rate kinematics (quadrotor.py:671-673) against the rotation they belong to: with the attitude rates the model returns,
#      the rotation matrix R(phi, theta, psi) = Rz Ry Rx (transformations.py:168-182, pinned to SciPy's extrinsic x-y-z in
#      tests/test_reference_pins.py) must turn with the body rates, dR/dt = R skew(omega) -- a property of the convention, not of the
#      matrix the reference wrote down (which _dynamics, the executed pin of the other rigid-body terms, does not state)
def test_euler_rate_kinematics_turn_the_rotation_with_the_body_rates():
    rng = np.random.default_rng(11)
    h = 1e-6
    for _ in range(40):
        x = np.zeros(12)
        x[6:9] = rng.uniform(-1.2, 1.2, size=3)
        x[9:12] = rng.uniform(-6, 6, size=3)
        hover = [P.m * P.g / 4] * 4
        ang_dot = np.array(rm.f_quad3d(list(x), hover, P))[6:9]
        Rp = np.array(rm.rot_xyz(*(x[6:9] + h * ang_dot)))
        Rm = np.array(rm.rot_xyz(*(x[6:9] - h * ang_dot)))
        R = np.array(rm.rot_xyz(*x[6:9]))
        w = x[9:12]
        skew = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        assert np.abs((Rp - Rm) / (2 * h) - R @ skew).max() < 1e-7       # central difference: O(h^2) + rounding / h


# ---- independent derivation (sympy matrix form) vs the hand-expanded oracle
@pytest.mark.parametrize('system', [2, 3, 'cartpole'])
def test_independent_derivation_agrees(system):
    rng = np.random.default_rng(3)
    p = rm.CartPoleParams() if system == 'cartpole' else P
    f_ind = ind.f_independent(system)
    pv = ind.params_vector(system, p)
    for _ in range(50):
        x = rng.normal(size=rm.NX[system]) * 0.4
        u = rng.uniform(-3, 3, size=1) if system == 'cartpole' else rng.uniform(0.03, 0.14, size=rm.NU[system])
        a = np.array(rm.F_BY_SYSTEM[system](list(x), list(u), p))
        b = np.asarray(f_ind(x, u, pv), dtype=float).ravel()
        assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)) < 1e-13


def test_rk4_close_to_exact_flow():
    """RK4 at dt = 1/60 vs DOP853 (rtol 1e-13), the CVODES stand-in: ~1e-9 at mild states."""
    rng = np.random.default_rng(5)
    worst = 0.0
    for _ in range(10):
        x = np.concatenate([rng.normal(size=6) * 0.3, rng.normal(size=3) * 0.1, rng.normal(size=3) * 0.2])
        u = P.m * P.g / 4 * (1 + 0.02 * rng.normal(size=4))
        a = np.array(rm.rk4_step(rm.f_quad3d, list(x), list(u), P, DT))
        b = ind.exact_step(3, x, u, P, DT)
        worst = max(worst, np.max(np.abs(a - b)))
    assert worst < 5e-8


def test_motor_map_is_identity_inside_pwm_range_and_clips_outside():
    lo, hi = rm.action_bounds(4)
    for T in np.linspace(lo, hi, 7)[1:-1]:
        assert rm.motor_map([T] * 4, P)[0] == pytest.approx(T, rel=1e-12)
    # the float32-rounded a_high sits 2.5e-8 above the true PWM_max thrust: the motor map clips it there
    assert rm.motor_map([hi] * 4, P)[0] == pytest.approx(P.kf * (P.pwm2rpm_scale * P.max_pwm + P.pwm2rpm_const) ** 2, rel=1e-12)
    assert rm.motor_map([0.0] * 4, P)[0] == pytest.approx(P.kf * (P.pwm2rpm_scale * P.min_pwm + P.pwm2rpm_const) ** 2, rel=1e-12)
    assert rm.motor_map([1.0] * 4, P)[0] == pytest.approx(P.kf * (P.pwm2rpm_scale * P.max_pwm + P.pwm2rpm_const) ** 2, rel=1e-12)
    # 2-D: each input drives two motors
    assert rm.motor_map([0.1, 0.2], P) == pytest.approx([0.1, 0.2], rel=1e-12)


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32-10."""
    f = lambda c, k: [int(v) for v in philox.philox4x32_10(np.array(c, np.uint32), np.array(k, np.uint32))]
    assert f([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert f([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert f([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_reset_offsets_are_uniform_in_range():
    lo, hi = np.full(12, -0.5), np.full(12, 0.5)
    draws = np.stack([philox.reset_offsets(1337, i, 7, lo, hi) for i in range(2000)])
    assert draws.min() >= -0.5 and draws.max() < 0.5
    assert abs(draws.mean()) < 0.01 and abs(draws.std() - 1 / math.sqrt(12)) < 0.01
    assert np.abs(np.corrcoef(draws.T) - np.eye(12)).max() < 0.08


def test_reference_trajectory_quirks():
    """3-D tracking reference: circle in 'zx' projected on the plane (point [0,0,.5], normal [0,1,1]);
    the velocity rows get the plane offset added too (transformations.py:119-122)."""
    pos, vel, speed = generate_trajectory('circle', 5, 1, 'zx', [0.5, 0], -0.5, DT)
    assert pos.shape == (300, 3) and speed.shape == (300, 1)
    assert pos[0] == pytest.approx([0, 0, 0]) and vel[0] == pytest.approx([-0.5 * 2 * math.pi / 5, 0, 0])
    M = projection_matrix([0, 0, 0.5], [0, 1, 1])
    n = np.array([0, 1, 1]) / math.sqrt(2)
    assert np.allclose(M[:3, :3], np.eye(3) - np.outer(n, n)) and np.allclose(M[:3, 3], (0.5 / math.sqrt(2)) * n)
    p2, v2 = transform_trajectory(pos, vel, [0, 0, 0.5], [0, 1, 1])
    assert np.allclose(v2[0], vel[0] @ (np.eye(3) - np.outer(n, n)) + M[:3, 3])
    env = RefEnv(system=3, task='traj_tracking', cost='quadratic', ctrl_freq=60, pyb_freq=240)
    assert env.X_GOAL.shape == (300, 12)
    assert np.allclose(env.X_GOAL[0, :6], [0, -0.62831853, 0.25, 0.25, 0.25, 0.25])


def test_numpy_batched_step_matches_python_oracle_and_c_port():
    """oracle/ref_batched.py (SURVEY 8d baseline (ii), the C4 step vectorised over envs) against the scalar oracle env by env,
    and against the C port over 150 steps with auto-resets (teacher-forced so that tumbling lanes cannot diverge)."""
    from oracle.c_port import CPort
    from oracle.ref_batched import BatchedC4
    from oracle.ref_env import INIT_STATE_LABELS
    rng = np.random.default_rng(21)
    ref = RefEnv(system=3, task='traj_tracking', cost='quadratic', ctrl_freq=60, pyb_freq=240, randomized_init=False,
                 reset_sampler=zeros_sampler)
    bat, port = BatchedC4(ref), CPort.from_ref_env(ref)
    lo = np.array([ref.INIT_STATE_RAND_INFO[k]['low'] for k in INIT_STATE_LABELS[3]])
    hi = np.array([ref.INIT_STATE_RAND_INFO[k]['high'] for k in INIT_STATE_LABELS[3]])
    n = 48
    x0 = rng.uniform(lo, hi, size=(n, 12))
    state, t = x0.copy(), rng.integers(0, 299, size=n).astype(np.int32)
    s0, t0 = state.copy(), t.copy()
    alo, ahi = ref.action_space_low, ref.action_space_high
    a = rng.uniform(alo - 0.1 * (ahi - alo), ahi + 0.1 * (ahi - alo), size=(n, 4))
    rew, done = bat.step(a, state, t, x0)
    for i in range(n):
        ref.reset()
        ref.state, ref.ctrl_step_counter = s0[i].copy(), int(t0[i])
        o, r, d, info = ref.step(a[i])
        assert bool(done[i]) == bool(d) and abs(r - rew[i]) <= 1e-12 * max(1, abs(r))
        assert np.max(np.abs(state[i] - (x0[i] if d else ref.state))) < 1e-12
    n = 512
    x0 = rng.uniform(lo, hi, size=(n, 12))
    sa, ta, sb, tb = x0.copy(), np.zeros(n, dtype=np.int32), x0.copy(), np.zeros(n, dtype=np.int32)
    dones = 0
    for k in range(150):
        a = rng.uniform(alo, ahi, size=(n, 4))
        ra, da, _, _ = port.step(a, sa, ta, x0, num_threads=2)
        rb, db = bat.step(a, sb, tb, x0)
        assert np.array_equal(da.astype(bool), db) and np.array_equal(ta, tb)
        assert np.abs(sa - sb).max() < 1e-11 and np.max(np.abs(ra - rb) / np.maximum(np.abs(ra), 1)) < 1e-11
        sb[...] = sa
        dones += int(db.sum())
    assert dones > n


def test_c_port_matches_python_oracle():
    from oracle.c_port import CPort
    rng = np.random.default_rng(11)
    cases = [
        (3, dict(task='traj_tracking', cost='quadratic', ctrl_freq=60, pyb_freq=240)),
        (2, dict(task='stabilization', cost='rl_reward', ctrl_freq=60, pyb_freq=240,
                 constraints=[{'constraint_form': 'bounded_constraint', 'constrained_variable': 'state',
                               'active_dims': [0, 2, 4], 'lower_bounds': [-2, 0, -0.5], 'upper_bounds': [2, 2, 0.5]}])),
        (1, dict(task='stabilization', cost='quadratic', ctrl_freq=50, pyb_freq=50)),
        ('cartpole', dict(task='traj_tracking', cost='rl_reward', ctrl_freq=50, pyb_freq=50)),
    ]
    for system, kw in cases:
        ref = RefEnv(system=system, randomized_init=False, reset_sampler=zeros_sampler, **kw)
        port = CPort.from_ref_env(ref)
        n, nx, nu = 64, ref.nx, ref.nu
        x0 = rng.normal(size=(n, nx)) * 0.15
        if system != 'cartpole':
            x0[:, {1: 0, 2: 2, 3: 4}[system]] += 1.0
        state, t = x0.copy(), rng.integers(0, 5, size=n).astype(np.int32)
        t0 = t.copy()
        lo, hi = ref.action_space_low, ref.action_space_high
        a = rng.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), size=(n, nu))
        rew, done, cv, viol = port.step(a, state, t, x0, want_constraints=ref.constraints is not None)
        for i in range(n):
            ref.reset()
            ref.state, ref.ctrl_step_counter = x0[i].copy(), int(t0[i])
            o, r, d, info = ref.step(a[i])
            assert bool(done[i]) == bool(d)
            assert abs(r - rew[i]) <= 1e-12 * max(1, abs(r))
            expect = x0[i] if d else ref.state
            assert np.max(np.abs(state[i] - expect)) < 1e-12
            if ref.constraints is not None:
                assert np.max(np.abs(cv[i] - info['constraint_values'])) < 1e-7      # oracle rounds to 8 decimals
                assert viol[i] == info['constraint_violation']


def test_golden_matches_oracle():
    """The committed fixtures are exactly what the oracle produces today (freezes the oracle)."""
    meta = json.load(open(os.path.join(HERE, 'golden', 'cases.json')))
    for name, case in meta['cases'].items():
        g = np.load(os.path.join(HERE, 'golden', name + '.npz'))
        system = case['system']
        sampler = PhiloxSampler(meta['seed'], system)
        refs = [RefEnv(system=system, env_index=i, reset_sampler=sampler, randomized_init=True, **case['kw'])
                for i in range(meta['n'])]
        vec = RefVecEnv(refs)
        sampler.serial = 0
        obs0, _ = vec.reset()
        assert np.array_equal(obs0, g['obs0'])
        for k in range(meta['t']):
            sampler.serial = k + 1
            o, r, d, info = vec.step(g['actions'][k].astype(np.float64))
            assert np.array_equal(o, g['obs'][k]) and np.array_equal(r, g['reward'][k])
            assert np.array_equal(d.astype(np.uint8), g['done'][k])


def test_aero_terms_known_answers():
    """Ground effect and drag of base_aviary.py:437-500 as restated for the analytic model (oracle.ref_model):
    level hover at height h gains thrust by GND_EFF_COEFF (PROP_RADIUS / 4h)^2; the clip height follows :145-147;
    drag at level attitude is k * v with k = -DRAG_COEFF * sum(2 pi rpm / 60); tilting by 90 deg about y maps a
    pure x velocity to a world force along -x again (R R = Ry(180 deg) flips x and z, and k_xy acts on it)."""
    from oracle import ref_model as rm
    p = rm.QuadParams()
    clip = rm.gnd_eff_h_clip()
    assert clip == pytest.approx(0.25 * 2.31348e-2 * np.sqrt(15 * 11.36859 / 4), rel=1e-12)
    hover = [p.m * p.g / 4] * 4
    x = np.zeros(12); x[4] = 0.1
    ug = rm.ground_effect_thrusts(3, x, hover, p)
    assert np.allclose(ug, np.array(hover) * (1 + 11.36859 * (2.31348e-2 / 0.4) ** 2), rtol=1e-14)
    x[4] = 0.001                                                      # below the clip height
    assert np.allclose(rm.ground_effect_thrusts(3, x, hover, p), np.array(hover) * (1 + 11.36859 * (2.31348e-2 / (4 * clip)) ** 2))
    x[4], x[6] = 0.1, 1.6                                             # |roll| >= pi/2: off (:463)
    assert rm.ground_effect_thrusts(3, x, hover, p) == hover
    x = np.zeros(12); x[4] = 0.05; x[7] = 0.2                         # pitched: front props (x > 0) are lower
    ug = rm.ground_effect_thrusts(3, x, hover, p)
    assert ug[0] > ug[1] and ug[3] > ug[2] and ug[0] == pytest.approx(ug[3]) and ug[1] == pytest.approx(ug[2])
    # drag
    x = np.zeros(12); x[1], x[3], x[5] = 1.0, -2.0, 0.5
    srpm = 4 * np.sqrt(hover[0] / p.kf)
    f = rm.drag_force(3, x, hover, p)
    want = [-9.1785e-7 * 2 * np.pi / 60 * srpm * 1.0, -9.1785e-7 * 2 * np.pi / 60 * srpm * -2.0, -10.311e-7 * 2 * np.pi / 60 * srpm * 0.5]
    assert np.allclose(f, want, rtol=1e-13)
    x = np.zeros(12); x[1] = 1.0; x[7] = np.pi / 2
    f = rm.drag_force(3, x, hover, p)
    assert f[0] == pytest.approx(9.1785e-7 * 2 * np.pi / 60 * srpm, rel=1e-12) and abs(f[2]) < 1e-18
    # the 2-D and 1-D reductions agree with the 3-D model restricted to their motion
    x2 = [0.3, 1.0, 0.2, 0.5, 0.3, 0.0]
    x3 = np.zeros(12); x3[0], x3[1], x3[4], x3[5], x3[7] = 0.3, 1.0, 0.2, 0.5, 0.3
    T = [2 * hover[0] * 1.1, 2 * hover[0] * 0.9]                      # T1 <-> motors 0, 3 (x > 0); T2 <-> motors 1, 2
    u3 = [T[0] / 2, T[1] / 2, T[1] / 2, T[0] / 2]
    f2, f3 = rm.drag_force(2, x2, T, p), rm.drag_force(3, x3, u3, p)
    assert np.allclose(f2, f3, rtol=1e-12, atol=1e-18)
    g2, g3 = rm.ground_effect_thrusts(2, x2, T, p), rm.ground_effect_thrusts(3, x3, u3, p)
    assert g2[0] == pytest.approx(g3[0] + g3[3], rel=1e-12) and g2[1] == pytest.approx(g3[1] + g3[2], rel=1e-12)


def test_golden_models_match_oracle():
    """tests/golden/models.npz (made by tests/golden/make_golden_models.py) freezes the oracle's model queries,
    loss and normalisers: regenerate-and-compare."""
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location('make_golden_models', os.path.join(here, 'golden', 'make_golden_models.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fresh = mod.build()
    frozen = np.load(os.path.join(here, 'golden', 'models.npz'))
    assert set(frozen.files) == set(fresh)
    for k in frozen.files:
        assert np.allclose(frozen[k], fresh[k], rtol=1e-13, atol=1e-15), k
