"""batched prior-model queries (df_func Jacobians, discretisation, quadratic loss).

CPU part: the oracle (sympy differentiation of the reference's matrix-form dynamics, oracle/ref_linearize.py)
against derivation-free central differences of the scalar restatement and against analytic known answers.
GPU part: the HIP kernels (hand-derived closed forms, csrc/scg_linearize.hip) against the oracle.
The reference holds no golden vectors for these functions: the pins are the
finite-difference cross-check and the hover / double-integrator known answers below.
"""
import numpy as np
import pytest

from oracle import ref_linearize, ref_model

SYSTEMS = [1, 2, 3, 'cartpole']
F = {1: ref_model.f_quad1d, 2: ref_model.f_quad2d, 3: ref_model.f_quad3d, 'cartpole': ref_model.f_cartpole}
RTOL = 1e-5          # north_star: 1e-5 relative, fp32


def params_of(system, inertial=None):
    if system == 'cartpole':
        return ref_model.CartPoleParams(*(inertial[:3] if inertial is not None else ()))
    return ref_model.QuadParams(*(inertial if inertial is not None else ()))


def sample(system, n, rng):
    nx, nu = ref_model.NX[system], ref_model.NU[system]
    x = rng.normal(size=(n, nx)) * 0.6
    if system == 'cartpole':
        u = rng.uniform(-10, 10, size=(n, nu))
    else:
        lo, hi = ref_model.action_bounds(nu)
        u = rng.uniform(lo, hi, size=(n, nu))
    return x, u


@pytest.mark.parametrize('system', SYSTEMS)
def test_oracle_jacobians_match_central_differences(system):
    rng = np.random.default_rng(7)
    p = params_of(system)
    xs, us = sample(system, 8, rng)
    nx, nu = xs.shape[1], us.shape[1]
    eps = 1e-6
    for x, u in zip(xs, us):
        A, B = ref_linearize.df(system, x, u, p)
        f = lambda a, b: np.asarray(F[system](a, b, p), dtype=np.float64)
        An = np.stack([(f(x + eps * np.eye(nx)[j], u) - f(x - eps * np.eye(nx)[j], u)) / (2 * eps) for j in range(nx)], 1)
        Bn = np.stack([(f(x, u + eps * np.eye(nu)[j]) - f(x, u - eps * np.eye(nu)[j])) / (2 * eps) for j in range(nu)], 1)
        scale = max(1.0, np.abs(A).max())
        assert np.abs(A - An).max() < 2e-7 * scale
        assert np.abs(B - Bn).max() < 2e-7 * max(1.0, np.abs(B).max())


def test_oracle_hover_known_answers():
    """Level hover, u = m g / 4 each: d ax / d theta = g, d ay / d phi = -g, d az / d f_i = 1 / m, and the torque
    rows are the mixer (quadrotor.py:667-669) divided by the inertias."""
    p = params_of(3)
    A, B = ref_linearize.df(3, np.zeros(12), np.full(4, p.m * p.g / 4), p)
    assert A[1, 7] == pytest.approx(p.g, rel=1e-15) and A[3, 6] == pytest.approx(-p.g, rel=1e-15)
    assert all(A[2 * k, 2 * k + 1] == 1.0 for k in range(3))
    assert A[6, 9] == 1.0 and A[7, 10] == 1.0 and A[8, 11] == 1.0
    expected_nonzero = {(0, 1), (2, 3), (4, 5), (1, 7), (3, 6), (6, 9), (7, 10), (8, 11)}
    assert {(a, b) for a in range(12) for b in range(12) if A[a, b] != 0.0} == expected_nonzero
    assert np.allclose(B[5], 1.0 / p.m, rtol=1e-15)
    k = p.l / np.sqrt(2.0)
    assert np.allclose(B[9], np.array([1, 1, -1, -1]) * k / p.ixx, rtol=1e-14)
    assert np.allclose(B[10], np.array([-1, 1, 1, -1]) * k / p.iyy, rtol=1e-14)
    assert np.allclose(B[11], np.array([1, -1, 1, -1]) * (p.km / p.kf) / p.izz, rtol=1e-14)


def test_oracle_discretisation_known_answer():
    """1-D quadrotor = double integrator: exact ZOH is Ad = [[1, dt], [0, 1]], Bd = [dt^2 / 2m, dt / m]; the
    reference's default (exact=False, mpc_utils.py:51-54) is the Euler form I + A dt, B dt."""
    p = params_of(1)
    A, B = ref_linearize.df(1, [1.0, 0.2], [0.3], p)
    dt = 1 / 60
    Ad, Bd = ref_linearize.discretize_linear_system(A, B, dt, exact=True)
    assert np.allclose(Ad, [[1, dt], [0, 1]], atol=1e-16) and np.allclose(Bd[:, 0], [dt * dt / 2 / p.m, dt / p.m], rtol=1e-13)
    Ae, Be = ref_linearize.discretize_linear_system(A, B, dt)
    assert np.array_equal(Ae, np.eye(2) + A * dt) and np.array_equal(Be, B * dt)


def test_oracle_loss_gradients_match_differences():
    rng = np.random.default_rng(3)
    Q, R = rng.normal(size=(6, 6)), rng.normal(size=(2, 2))          # deliberately not symmetric
    x, u, xr, ur = rng.normal(size=6), rng.normal(size=2), rng.normal(size=6), rng.normal(size=2)
    out = ref_linearize.quadratic_loss(x, u, xr, ur, Q, R)
    eps = 1e-6
    gx = [(ref_linearize.quadratic_loss(x + eps * np.eye(6)[j], u, xr, ur, Q, R)['l']
           - ref_linearize.quadratic_loss(x - eps * np.eye(6)[j], u, xr, ur, Q, R)['l']) / (2 * eps) for j in range(6)]
    gu = [(ref_linearize.quadratic_loss(x, u + eps * np.eye(2)[j], xr, ur, Q, R)['l']
           - ref_linearize.quadratic_loss(x, u - eps * np.eye(2)[j], xr, ur, Q, R)['l']) / (2 * eps) for j in range(2)]
    assert np.allclose(out['l_x'], gx, atol=1e-8) and np.allclose(out['l_u'], gu, atol=1e-8)


def test_host_model_matches_oracle():
    """env.symbolic for host arrays (safe_control_gym_amd.math_and_models.symbolic_systems) vs the oracle."""
    from safe_control_gym_amd.math_and_models.symbolic_systems import AnalyticModel
    p = params_of(3)
    consts = dict(m=p.m, g=p.g, l=p.l, Ixx=p.ixx, Iyy=p.iyy, Izz=p.izz, gamma=p.km / p.kf)
    model = AnalyticModel('quad3d', consts, 1 / 60)
    rng = np.random.default_rng(11)
    xs, us = sample(3, 4, rng)
    out = model.df_func(xs, us)
    for k in range(4):
        A, B = ref_linearize.df(3, xs[k], us[k], p)
        assert np.allclose(out['dfdx'][k], A, rtol=1e-12, atol=1e-12) and np.allclose(out['dfdu'][k], B, rtol=1e-12)


# ------------------------------------------------------------------------------------------------------
# GPU: the HIP kernels against the oracle
# ------------------------------------------------------------------------------------------------------
def _make(system, n, **kw):
    import safe_control_gym_amd as scg
    if system == 'cartpole':
        return scg.make('cartpole', num_envs=n, device='cuda', seed=1, **kw)
    if system == 3:
        kw.setdefault('task_info', {'stabilization_goal': [0.0, 0.0, 1.0], 'stabilization_goal_tolerance': 0.05})
    return scg.make('quadrotor', quad_type=system, num_envs=n, device='cuda', seed=1, **kw)


def _check(A_dev, B_dev, xs, us, system, params, disc_dt=0.0, idx=None):
    idx = range(len(xs)) if idx is None else idx
    worst = 0.0
    for k in idx:
        A, B = ref_linearize.df(system, xs[k], us[k], params[k] if isinstance(params, list) else params)
        if disc_dt:
            A, B = ref_linearize.discretize_linear_system(A, B, disc_dt)
        for dev, ref in ((A_dev[k], A), (B_dev[k], B)):
            # structural zeros are exact; everything else within 1e-5 of the entry (or of 1 for small entries)
            assert np.array_equal(dev[ref == 0.0], ref[ref == 0.0]), (k, dev, ref)
            err = np.abs(dev - ref) / np.maximum(np.abs(ref), 1.0)
            worst = max(worst, err.max())
    assert worst <= RTOL, worst
    return worst


@pytest.mark.gpu
@pytest.mark.parametrize('system', SYSTEMS)
@pytest.mark.parametrize('n', [1, 63, 65, 1000])
def test_linearize_kernel_matches_oracle(gpu, system, n):
    import torch
    env = _make(system, n)
    env.reset()
    rng = np.random.default_rng(100 + n)
    xs, us = sample(system, n, rng)
    A, B = env.linearize(torch.as_tensor(xs, dtype=torch.float32, device='cuda'),
                         torch.as_tensor(us, dtype=torch.float32, device='cuda'))
    assert A.shape == (n, xs.shape[1], xs.shape[1]) and B.shape == (n, xs.shape[1], us.shape[1])
    x32, u32 = xs.astype(np.float32).astype(np.float64), us.astype(np.float32).astype(np.float64)
    _check(A.cpu().numpy().astype(np.float64), B.cpu().numpy().astype(np.float64), x32, u32, system, params_of(system))
    # discrete variant (discretize_linear_system, exact=False)
    Ad, Bd = env.linearize(torch.as_tensor(xs, dtype=torch.float32, device='cuda'),
                           torch.as_tensor(us, dtype=torch.float32, device='cuda'), discrete=True)
    _check(Ad.cpu().numpy().astype(np.float64), Bd.cpu().numpy().astype(np.float64), x32, u32, system,
           params_of(system), disc_dt=float(np.float32(env.CTRL_TIMESTEP)))
    env.close()


@pytest.mark.gpu
def test_linearize_defaults_current_state_and_symbolic_api(gpu):
    """env.linearize() about the envs' own states (zero copy) and U_EQ; env.symbolic.df_func on device tensors."""
    import torch
    n = 300
    env = _make(3, n, randomized_init=True)
    env.reset()
    for _ in range(5):
        env.step(torch.full((n, 4), 0.09, device='cuda'))
    xs = env.state.cpu().numpy().astype(np.float64)
    us = np.tile(np.asarray(env.U_EQ, dtype=np.float32).astype(np.float64), (n, 1))
    A, B = env.linearize()
    _check(A.cpu().numpy().astype(np.float64), B.cpu().numpy().astype(np.float64), xs, us, 3, params_of(3))
    out = env.symbolic.df_func(env.state, torch.as_tensor(us, dtype=torch.float32, device='cuda'))
    assert torch.equal(out['dfdx'], A) and torch.equal(out['dfdu'], B)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('system', [3, 'cartpole'])
def test_linearize_uses_per_env_inertial_properties(gpu, system):
    import torch
    n = 130
    env = _make(system, n, randomized_inertial_prop=True)
    env.reset()
    prop = env.inertial_properties.cpu().numpy().astype(np.float64)          # [N, 4]
    assert np.ptp(prop[:, 0]) > 0
    rng = np.random.default_rng(5)
    xs, us = sample(system, n, rng)
    env.state[:] = torch.as_tensor(xs, dtype=torch.float32, device='cuda')
    A, B = env.linearize(input=torch.as_tensor(us, dtype=torch.float32, device='cuda'))
    params = [params_of(system, list(prop[k])) for k in range(n)]
    _check(A.cpu().numpy().astype(np.float64), B.cpu().numpy().astype(np.float64),
           xs.astype(np.float32).astype(np.float64), us.astype(np.float32).astype(np.float64), system, params)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('system', SYSTEMS)
def test_quadratic_loss_kernel_matches_oracle(gpu, system):
    import torch
    n = 257
    env = _make(system, n, cost='quadratic')
    env.reset()
    rng = np.random.default_rng(9)
    xs, us = sample(system, n, rng)
    nx, nu = xs.shape[1], us.shape[1]
    Q, R = rng.normal(size=(nx, nx)), rng.normal(size=(nu, nu))          # general (non-symmetric) weights
    xr_rows, ur_vec = rng.normal(size=(n, nx)), rng.normal(size=nu)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device='cuda')
    out = env.quadratic_loss(t(xs), t(us), x_ref=t(xr_rows), u_ref=ur_vec, Q=Q, R=R)
    out2 = env.symbolic.loss(t(xs), t(us), t(xr_rows[0]), ur_vec, Q, R)      # CasADi-style call, one reference for all
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    Qf, Rf = f32(Q), f32(R)
    for k in range(n):
        ref = ref_linearize.quadratic_loss(f32(xs[k]), f32(us[k]), f32(xr_rows[k]), f32(ur_vec), Qf, Rf)
        scale = max(1.0, abs(ref['l']))
        assert abs(out['l'][k].item() - ref['l']) <= 2e-5 * scale          # sums of ~150 products with mixed signs
        assert np.abs(out['l_x'][k].cpu().numpy() - ref['l_x']).max() <= 2e-5 * max(1.0, np.abs(ref['l_x']).max())
        assert np.abs(out['l_u'][k].cpu().numpy() - ref['l_u']).max() <= 2e-5 * max(1.0, np.abs(ref['l_u']).max())
        ref2 = ref_linearize.quadratic_loss(f32(xs[k]), f32(us[k]), f32(xr_rows[0]), f32(ur_vec), Qf, Rf)
        assert abs(out2['l'][k].item() - ref2['l']) <= 2e-5 * max(1.0, abs(ref2['l']))
    assert np.allclose(out['l_xx'][3].cpu().numpy(), 0.5 * (Qf + Qf.T), rtol=1e-6)
    assert out['l_xu'].shape == (n, nx, nu) and out['l_uu'].shape == (n, nu, nu)
    # the env's own reward under cost=quadratic is -l at the stabilisation goal with the env's Q, R (quadrotor.py:916-930)
    env.close()


@pytest.mark.gpu
def test_linearize_full_size_properties(gpu):
    """2^20 envs (BASELINE config C4 size): structure, finiteness, a sampled oracle comparison, and
    Ad - I == dt * A (the discretisation is linear in A)."""
    import torch
    n = 1 << 20
    env = _make(3, n, randomized_init=True)
    env.reset()
    A, B = env.linearize()
    assert torch.isfinite(A).all() and torch.isfinite(B).all()
    zero_mask = torch.ones(12, 12, dtype=torch.bool, device='cuda')
    for a, b in [(0, 1), (2, 3), (4, 5), (1, 6), (1, 7), (1, 8), (3, 6), (3, 7), (3, 8), (5, 6), (5, 7), (6, 6), (6, 7),
                 (6, 9), (6, 10), (6, 11), (7, 6), (7, 10), (7, 11), (8, 6), (8, 7), (8, 10), (8, 11), (9, 10), (9, 11),
                 (10, 9), (10, 11), (11, 9), (11, 10)]:
        zero_mask[a, b] = False
    assert (A[:, zero_mask] == 0).all()
    assert (A[:, 0, 1] == 1).all() and (A[:, 6, 9] == 1).all()
    idx = np.random.default_rng(0).choice(n, 128, replace=False)
    xs = env.state[torch.as_tensor(idx, device='cuda')].cpu().numpy().astype(np.float64)
    us = np.tile(np.asarray(env.U_EQ, dtype=np.float32).astype(np.float64), (128, 1))
    _check(A[torch.as_tensor(idx, device='cuda')].cpu().numpy().astype(np.float64),
           B[torch.as_tensor(idx, device='cuda')].cpu().numpy().astype(np.float64), xs, us, 3, params_of(3))
    Ad, Bd = env.linearize(discrete=True)
    dt = np.float32(env.CTRL_TIMESTEP)
    eye = torch.eye(12, device='cuda')
    assert torch.allclose(Ad - eye, A * float(dt), rtol=1e-6, atol=1e-7) and torch.allclose(Bd, B * float(dt), rtol=1e-6)
    env.close()


@pytest.mark.gpu
def test_linearize_c_abi_rejects_bad_arguments(gpu):
    import torch
    from safe_control_gym_amd import _capi
    env = _make(2, 8)
    env.reset()
    eng, lib = env._engine, _capi.load()
    A = torch.empty((8, 6, 6), device='cuda')
    B = torch.empty((8, 6, 2), device='cuda')
    p = _capi.ptr
    st, ac = env._bufs['state'], env._bufs['action']
    assert lib.scg_linearize(eng.handle, None, p(ac), None, p(A), p(B), 0.0, env._ld, 8, None) < 0
    assert lib.scg_linearize(eng.handle, p(st), p(ac), None, p(A), p(B), -1.0, env._ld, 8, None) < 0
    assert lib.scg_linearize(eng.handle, p(st), p(ac), None, p(A), p(B), 0.0, 4, 8, None) < 0          # ld < n
    assert lib.scg_linearize(eng.handle, p(st), p(ac), None, p(A), p(B), 0.0, env._ld, 0, None) == 0   # empty batch
    assert b'ld' in lib.scg_last_error() or lib.scg_last_error() is not None
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('system', SYSTEMS)
def test_model_eval_matches_oracle(gpu, system):
    """env.symbolic.fc_func / rk4_func on device tensors: x_dot = f(x, u) (symbolic_systems.py:68) and one zero-order-hold step of
    the env's integrator (rk_discrete, mpc_utils.py:79-101) for explicit (x, u) pairs, without touching the envs."""
    import torch
    n = 333
    env = _make(system, n, ctrl_freq=50, pyb_freq=50)
    env.reset()
    before = env.state.clone()
    rng = np.random.default_rng(21)
    xs, us = sample(system, n, rng)
    xt = torch.as_tensor(xs, dtype=torch.float32, device='cuda')
    ut = torch.as_tensor(us, dtype=torch.float32, device='cuda')
    f_dev = env.symbolic.fc_func(xt, ut)['f'].cpu().numpy().astype(np.float64)
    x_dev = env.symbolic.rk4_func(xt, ut)['xf'].cpu().numpy().astype(np.float64)        # (fd_func, the error-controlled flow: tests/test_model_flow.py)
    assert torch.equal(env.state, before)                                  # the envs were not advanced
    p = params_of(system)
    x32, u32 = xs.astype(np.float32).astype(np.float64), us.astype(np.float32).astype(np.float64)
    dt = env.CTRL_TIMESTEP
    for k in range(n):
        f_ref = np.asarray(F[system](list(x32[k]), list(u32[k]), p))
        x_ref = np.asarray(ref_model.rk4_step(F[system], list(x32[k]), list(u32[k]), p, dt))
        # f is a sum of terms that cancel (torque rows: 2005 N^-1 s^-2 times thrust differences; T/m against g):
        # the 1e-5 bound is relative to the size of those terms, |A||x| + |B||u| (+ gravity), not to the cancelled sum
        A, B = ref_linearize.df(system, x32[k], u32[k], p)
        scale = 1.0 + np.abs(A) @ np.abs(x32[k]) + np.abs(B) @ np.abs(u32[k]) + p.g
        assert np.max(np.abs(f_dev[k] - f_ref) / scale) <= RTOL, (k, f_dev[k], f_ref)
        assert np.max(np.abs(x_dev[k] - x_ref) / np.maximum(np.abs(x_ref), 1.0 + dt * scale)) <= RTOL, (k, x_dev[k], x_ref)
    # consistency with the env itself: stepping the envs from these states with the same (in-bounds) inputs
    if system != 'cartpole':
        env2 = _make(system, n, ctrl_freq=50, pyb_freq=50, done_on_out_of_bound=False, auto_reset=False)
        env2.reset()
        env2.state[:] = xt
        env2.step(ut)
        # same integrator; step() additionally sends the thrusts through the motor map round trip (quadrotor_utils.py:21-67)
        assert torch.allclose(env2.state, env.model_step(xt, ut), rtol=1e-5, atol=2e-5)
        env2.close()
    env.close()


@pytest.mark.gpu
def test_golden_models_on_device(gpu):
    """Replay tests/golden/models.npz (frozen oracle outputs) through the HIP kernels: df_func, fc_func, fd_func, the
    quadratic loss with its gradients, and both normalisers."""
    import os
    import torch
    from safe_control_gym_amd.math_and_models.normalization import MeanStdNormalizer, RewardStdNormalizer
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'models.npz'))
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device='cuda')
    rel = lambda dev, ref, scale=1.0: float(np.max(np.abs(dev.cpu().numpy().astype(np.float64) - ref) / np.maximum(np.abs(ref), scale)))
    for tag, system in (('q1', 1), ('q2', 2), ('q3', 3), ('cp', 'cartpole')):
        env = _make(system, 16, ctrl_freq=50, pyb_freq=50, cost='quadratic')
        env.reset()
        x, u = t(g[tag + '_x']), t(g[tag + '_u'])
        A, B = env.linearize(x, u)
        assert rel(A, g[tag + '_A']) <= RTOL and rel(B, g[tag + '_B']) <= RTOL
        # f cancels large terms (see test_model_eval_matches_oracle): bound relative to |A||x| + |B||u| + g
        scale = 1.0 + np.abs(g[tag + '_A']) @ np.abs(g[tag + '_x'])[..., None] + np.abs(g[tag + '_B']) @ np.abs(g[tag + '_u'])[..., None] + 9.8
        f = env.model_dynamics(x, u).cpu().numpy().astype(np.float64)
        assert np.max(np.abs(f - g[tag + '_f']) / scale[..., 0]) <= RTOL
        xn = env.model_step(x, u).cpu().numpy().astype(np.float64)
        assert np.max(np.abs(xn - g[tag + '_xnext']) / np.maximum(np.abs(g[tag + '_xnext']), 1.0 + 0.02 * scale[..., 0])) <= RTOL
        out = env.quadratic_loss(x, u, x_ref=g[tag + '_xr'], u_ref=g[tag + '_ur'], Q=g[tag + '_Q'], R=g[tag + '_R'])
        assert rel(out['l'], g[tag + '_l'], max(1.0, np.abs(g[tag + '_l']).max())) <= 2e-5
        assert rel(out['l_x'], g[tag + '_lx'], max(1.0, np.abs(g[tag + '_lx']).max())) <= 2e-5
        assert rel(out['l_u'], g[tag + '_lu'], max(1.0, np.abs(g[tag + '_lu']).max())) <= 2e-5
        env.close()
    norm = MeanStdNormalizer(shape=(5,), clip=3.0)
    for b in range(3):
        y = norm(t(g['norm_x%d' % b]))
        assert rel(y, g['norm_y%d' % b]) <= 5 * RTOL
        assert np.allclose(norm.rms.mean.cpu().numpy(), g['norm_mean%d' % b], rtol=1e-12, atol=1e-12)
        assert np.allclose(norm.rms.var.cpu().numpy(), g['norm_var%d' % b], rtol=1e-10)
    rnorm = RewardStdNormalizer(gamma=0.9, clip=5.0)
    for b in range(6):
        out = rnorm(t(g['rnorm_rew%d' % b]), torch.as_tensor(g['rnorm_done%d' % b], device='cuda'))
        assert rel(out, g['rnorm_out%d' % b]) <= 5 * RTOL
        assert rel(rnorm.ret, g['rnorm_ret%d' % b]) <= 2e-5
        assert float(rnorm.rms.var) == pytest.approx(float(g['rnorm_var%d' % b]), rel=1e-5)


# ---- exact discretisation (mpc_utils.py:45-50) and the linearised-model functions (symbolic_systems.py:93-112) -----------
def test_oracle_fd_linear_closed_form_matches_numerical_integration():
    """fd_linear_func integrates an AFFINE ode: the matrix-exponential flow the oracle (and the device) use equals a
    tight numerical integration of the same ode (DOP853, rtol 1e-12 -- the CVODES stand-in; CVODES itself is absent)."""
    rng = np.random.default_rng(3)
    for system in SYSTEMS:
        p = params_of(system)
        xs, us = sample(system, 3, rng)
        for x, u in zip(xs, us):
            xe, ue = x + 0.1 * rng.normal(size=x.shape), u * (1 + 0.05 * rng.normal(size=u.shape))
            a = ref_linearize.fd_linear(system, xe, ue, x, u, p, 1 / 60)
            b = ref_linearize.fd_linear(system, xe, ue, x, u, p, 1 / 60, method='ivp')
            assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)) < 1e-9
            # at the linearisation point the linear model's vector field is the model's
            assert np.allclose(ref_linearize.fc_linear(system, x, u, x, u, p), F[system](x, u, p), rtol=0, atol=1e-12)


def _rand_systems(rng, n, nx, m, norms):
    """n random (A, B) pairs with ||A_k||_inf = norms[k]; B entries of very different sizes (they enter expm linearly)."""
    A = rng.normal(size=(n, nx, nx))
    A *= (np.asarray(norms) / np.abs(A).sum(2).max(1))[:, None, None]
    B = rng.normal(size=(n, nx, m)) * rng.choice([0.1, 1.0, 300.0], size=(n, 1, 1))
    return A, B


@pytest.mark.gpu
@pytest.mark.parametrize('nx,m', [(12, 4), (6, 2), (4, 1), (2, 1), (12, 1), (5, 3), (15, 1), (1, 1)])
@pytest.mark.parametrize('n', [1, 63, 65])
def test_exact_discretisation_matches_expm(gpu, nx, m, n):
    """scg_discretize_exact (MFMA Taylor + scaling and squaring) against scipy.linalg.expm of the block matrix, the
    reference's own recipe (mpc_utils.py:45-50): small norms (no squaring), norms that need 1-6 squarings, every packing
    of envs per wave (16 / 8 / 4 rows per env) and ragged batch sizes; the optional z1 = Ad z0 + Bd w epilogue too."""
    import torch
    import scipy.linalg
    from safe_control_gym_amd.controllers.mpc.mpc_utils import discretize_exact, discretize_linear_system
    rng = np.random.default_rng(1000 * nx + 10 * m + n)
    norms = rng.choice([0.05, 0.4, 3.0, 25.0], size=n)           # ||A||_inf dt after scaling by dt below
    dt = 1 / 60
    A, B = _rand_systems(rng, n, nx, m, norms / dt)
    z0, w = rng.normal(size=(n, nx)), rng.normal(size=(n, m))
    dev = lambda a: torch.as_tensor(a, dtype=torch.float32, device='cuda')
    Ad, Bd, z1 = discretize_exact(dev(A), dev(B), dt, z0=dev(z0), w=dev(w))
    Ad, Bd, z1 = Ad.double().cpu().numpy(), Bd.double().cpu().numpy(), z1.double().cpu().numpy()
    A32, B32 = A.astype(np.float32).astype(np.float64), B.astype(np.float32).astype(np.float64)
    z32, w32 = z0.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64)
    for k in range(n):
        M = np.zeros((nx + m, nx + m))
        M[:nx, :nx], M[:nx, nx:] = A32[k], B32[k]
        E = scipy.linalg.expm(M * float(np.float32(dt)))
        ra, rb = E[:nx, :nx], E[:nx, nx:]
        # error relative to the size of the block (squaring amplifies rounding with the norm of exp(A dt))
        sa, sb = max(1.0, np.abs(ra).max()), max(1.0, np.abs(rb).max())
        tol = RTOL * (1.0 if norms[k] < 1.0 else 8.0)
        assert np.abs(Ad[k] - ra).max() <= tol * sa, (k, norms[k])
        assert np.abs(Bd[k] - rb).max() <= tol * sb, (k, norms[k])
        rz = ra @ z32[k] + rb @ w32[k]
        assert np.abs(z1[k] - rz).max() <= 4 * tol * max(1.0, np.abs(rz).max(), sa * np.abs(z32[k]).max(), sb * np.abs(w32[k]).max())
    # the reference's signature, single NumPy system in / NumPy out, and the exact=False branch
    a1, b1 = discretize_linear_system(A[0], B[0], dt, exact=True)
    # (a lone system is scaled by its own norm, a packed one by its wave's largest: equal up to rounding, not bit for bit)
    assert isinstance(a1, np.ndarray) and a1.shape == (nx, nx)
    assert np.abs(a1 - Ad[0]).max() <= 8 * RTOL * max(1.0, np.abs(Ad[0]).max()) and np.abs(b1 - Bd[0]).max() <= 8 * RTOL * max(1.0, np.abs(Bd[0]).max())
    a0, b0 = discretize_linear_system(A[0], B[0], dt)
    assert np.allclose(a0, np.eye(nx) + A[0] * dt) and np.allclose(b0, B[0] * dt)


@pytest.mark.gpu
def test_exact_discretisation_full_size_properties(gpu):
    """2^20 16 x 16 systems in one launch.  Size-independent properties of the matrix exponential: (i) a spot check of 256
    random systems against scipy; (ii) semigroup: discretising with dt and stepping twice equals discretising with 2 dt
    (Ad(2 dt) = Ad(dt)^2, Bd(2 dt) = (Ad(dt) + I) Bd(dt)); (iii) a nilpotent double integrator gives its polynomial exactly."""
    import torch
    import scipy.linalg
    from safe_control_gym_amd.controllers.mpc.mpc_utils import discretize_exact
    n, nx, m, dt = 1 << 20, 12, 4, 1 / 60
    g = torch.Generator(device='cuda'); g.manual_seed(4)
    A = torch.randn((n, nx, nx), generator=g, device='cuda') * 3.0
    B = torch.randn((n, nx, m), generator=g, device='cuda') * 30.0
    A[7] = 0.0; A[7, 0, 1] = 1.0; A[7, 2, 3] = 1.0                  # env 7: double integrators x'' = u0, y'' = u1
    B[7] = 0.0; B[7, 1, 0] = 1.0; B[7, 3, 1] = 1.0
    Ad, Bd, _ = discretize_exact(A, B, dt)
    Ad2, Bd2, _ = discretize_exact(A, B, 2 * dt)
    eye = torch.eye(nx, device='cuda')
    sq = torch.bmm(Ad, Ad)
    assert float(((sq - Ad2).abs().amax((1, 2)) / Ad2.abs().amax((1, 2)).clamp(min=1.0)).max()) <= 4 * RTOL
    comp = torch.bmm(Ad + eye, Bd)
    assert float(((comp - Bd2).abs().amax((1, 2)) / Bd2.abs().amax((1, 2)).clamp(min=1.0)).max()) <= 4 * RTOL
    h = float(np.float32(dt))
    assert torch.equal(Ad[7, 0, :4], torch.tensor([1.0, h, 0.0, 0.0], device='cuda'))
    assert abs(float(Bd[7, 0, 0]) - h * h / 2) <= 1e-9 and abs(float(Bd[7, 1, 0]) - h) <= 1e-9
    idx = np.random.default_rng(0).integers(0, n, size=256)
    An, Bn = A[idx].double().cpu().numpy(), B[idx].double().cpu().numpy()
    Adn, Bdn = Ad[idx].double().cpu().numpy(), Bd[idx].double().cpu().numpy()
    for k in range(256):
        M = np.zeros((16, 16)); M[:12, :12], M[:12, 12:] = An[k], Bn[k]
        E = scipy.linalg.expm(M * h)
        assert np.abs(Adn[k] - E[:12, :12]).max() <= 4 * RTOL * max(1.0, np.abs(E[:12, :12]).max())
        assert np.abs(Bdn[k] - E[:12, 12:]).max() <= 4 * RTOL * max(1.0, np.abs(E[:12, 12:]).max())


@pytest.mark.gpu
@pytest.mark.parametrize('system', SYSTEMS)
@pytest.mark.parametrize('n', [1, 63, 65])
def test_linear_model_queries_match_oracle(gpu, system, n):
    """env.linearize(discrete='exact'), env.model_linear_dynamics (fc_linear_func) and env.model_linear_step
    (fd_linear_func), plus their env.symbolic spellings, against the oracle (symbolic Jacobians + scipy expm)."""
    import torch
    env = _make(system, n)
    env.reset()
    rng = np.random.default_rng(500 + n)
    xs, us = sample(system, n, rng)
    if system == 3:
        xs[:, 6:9] = np.clip(xs[:, 6:9], -1.0, 1.0)          # away from the Euler-angle singularity: exp(A dt) of a matrix with
        # 1 / cos(theta) entries amplifies the float32 rounding of its INPUT beyond any fixed tolerance
    xe = xs + 0.1 * rng.normal(size=xs.shape)
    ue = us * (1 + 0.05 * rng.normal(size=us.shape))
    dev = lambda a: torch.as_tensor(a, dtype=torch.float32, device='cuda')
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    p = params_of(system)
    dt = float(np.float32(env.CTRL_TIMESTEP))
    Ad, Bd = env.linearize(dev(xs), dev(us), discrete='exact')
    fl = env.model_linear_dynamics(dev(xe), dev(ue), dev(xs), dev(us)).double().cpu().numpy()
    zl = env.model_linear_step(dev(xe), dev(ue), dev(xs), dev(us)).double().cpu().numpy()
    sym_f = env.symbolic.fc_linear_func(dev(xe), dev(ue), dev(xs), dev(us))['f_linear'].double().cpu().numpy()
    sym_z = env.symbolic.fd_linear_func(dev(xe), torch.cat([dev(ue), dev(xs), dev(us)], dim=1))['xf'].double().cpu().numpy()
    assert np.array_equal(sym_f, fl) and np.array_equal(sym_z, zl)
    assert torch.equal(env.symbolic.g_linear_func(dev(xe), dev(ue), dev(xs), dev(us))['g_linear'], dev(xs) + (dev(xe) - dev(xs)))
    Adn, Bdn = Ad.double().cpu().numpy(), Bd.double().cpu().numpy()
    worst = 0.0
    for k in range(n):
        A, B = ref_linearize.df(system, r32(xs)[k], r32(us)[k], p)
        ra, rb = ref_linearize.discretize_linear_system(A, B, dt, exact=True)
        worst = max(worst, np.abs(Adn[k] - ra).max() / max(1.0, np.abs(ra).max()), np.abs(Bdn[k] - rb).max() / max(1.0, np.abs(rb).max()))
        rf = ref_linearize.fc_linear(system, r32(xe)[k], r32(ue)[k], r32(xs)[k], r32(us)[k], p)
        rz = ref_linearize.fd_linear(system, r32(xe)[k], r32(ue)[k], r32(xs)[k], r32(us)[k], p, dt)
        # the linear model's field mixes terms of very different size (torque rows ~ 1e3 u): error relative to the largest
        # term that enters the row, as for any fp32 dot product
        scale_f = np.maximum(np.abs(rf), 1.0) + np.abs(A) @ np.abs(r32(xe)[k] - r32(xs)[k]) + np.abs(B) @ np.abs(r32(ue)[k] - r32(us)[k])
        worst = max(worst, (np.abs(fl[k] - rf) / scale_f).max(), (np.abs(zl[k] - rz) / np.maximum(np.abs(rz), 1.0)).max())
    assert worst <= 2 * RTOL, worst
    env.close()


@pytest.mark.gpu
def test_linear_model_full_size_properties(gpu):
    """2^20 envs: size-independent properties of fc_linear_func / fd_linear_func (symbolic_systems.py:93-112).
    (i) at the expansion point the linear model IS the model: f_lin(x*, u*; x*, u*) == f(x*, u*);
    (ii) affinity: f_lin(x* + a d) - f_lin(x*) == a (f_lin(x* + d) - f_lin(x*));
    (iii) fd_linear at (x*, u*) with f(x*, u*) = 0 (hover) is the fixed point x*; (iv) a sampled oracle comparison."""
    import torch
    n = 1 << 20
    env = _make(3, n, randomized_init=True)
    env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(9)
    xs = env.state.clone()
    xs[:, 6:9].clamp_(-1.0, 1.0)
    us = torch.as_tensor(np.asarray(env.U_EQ, dtype=np.float32), device='cuda').expand(n, -1) * \
        (1 + 0.1 * torch.randn((n, 4), generator=g, device='cuda'))
    f0 = env.model_linear_dynamics(xs, us, xs, us)
    f = env.symbolic.fc_func(xs, us)['f']
    assert float(((f0 - f).abs() / f.abs().clamp(min=1.0)).max()) <= RTOL
    d = 0.05 * torch.randn((n, 12), generator=g, device='cuda')
    f1 = env.model_linear_dynamics(xs + d, us, xs, us)
    f3 = env.model_linear_dynamics(xs + 3 * d, us, xs, us)
    scale = (f1 - f0).abs().amax(1, keepdim=True).clamp(min=1.0) + f0.abs()
    assert float((((f3 - f0) - 3 * (f1 - f0)).abs() / scale).max()) <= 8 * RTOL
    hover = torch.zeros((n, 12), device='cuda'); hover[:, 0:6:2] = xs[:, 0:6:2]
    ueq = torch.as_tensor(np.asarray(env.U_EQ, dtype=np.float32), device='cuda').expand(n, -1).contiguous()
    z = env.model_linear_step(hover, ueq, hover, ueq)
    assert float((z - hover).abs().max()) <= 1e-5
    idx = np.random.default_rng(1).choice(n, 64, replace=False)
    ti = torch.as_tensor(idx, device='cuda')
    p = params_of(3)
    r64 = lambda t: t[ti].double().cpu().numpy()
    xe = xs + d
    for k in range(64):
        rf = ref_linearize.fc_linear(3, r64(xe)[k], r64(us)[k], r64(xs)[k], r64(us)[k], p)
        assert (np.abs(r64(f1)[k] - rf) / np.maximum(np.abs(rf), 1.0)).max() <= 4 * RTOL
    env.close()
