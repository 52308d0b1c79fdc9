"""SymbolicModel.fd_func as the reference defines it (math_and_models/symbolic_systems.py:70-73: CasADi's CVODES integrator of the
prior model over [0, dt]) and its forward sensitivities: ``scg_model_flow`` (csrc/scg_flow.hip) and the NumPy mirror in
``AnalyticModel`` against the oracle's independent answer -- DOP853 at rtol 1e-13 on the sympy matrix-form dynamics and on their
variational equations (oracle/independent.py).

Parity unpinned by construction: CasADi / SUNDIALS do not import here and the reference holds no fixture of ``fd_func`` (nothing in
the reference calls it).  What is checked is the definition -- the flow of the ODE to the stated tolerance -- not CVODES' own
rounding inside that tolerance.
"""
import numpy as np
import pytest

from oracle import independent, ref_model

SYSTEMS = [1, 2, 3, 'cartpole']
NAMES = {1: 'quad1d', 2: 'quad2d', 3: 'quad3d', 'cartpole': 'cartpole'}
CVODES_RTOL, CVODES_ATOL = 1e-6, 1e-8          # CasADi's integrator defaults


def params_of(system, inertial=None):
    if system == 'cartpole':
        return ref_model.CartPoleParams(*(inertial[:3] if inertial is not None else ()))
    return ref_model.QuadParams(*(inertial if inertial is not None else ()))


def sample(system, n, rng):
    nx, nu = ref_model.NX[system], ref_model.NU[system]
    x = rng.normal(size=(n, nx)) * 0.6
    if system == 3:
        x[:, 6:9] = np.clip(x[:, 6:9], -1.2, 1.2)      # away from the Euler-angle singularity of the rate matrix
        x[:, 9:12] *= 3.0                              # body rates of a few rad/s
    if system == 2:
        x[:, 4] = np.clip(x[:, 4], -1.2, 1.2)
    if system == 'cartpole':
        u = rng.uniform(-10, 10, size=(n, nu))
    else:
        lo, hi = ref_model.action_bounds(nu)
        u = rng.uniform(lo, hi, size=(n, nu))
    return x, u


def _make(system, n, **kw):
    import safe_control_gym_amd as scg
    if system == 'cartpole':
        return scg.make('cartpole', num_envs=n, device='cuda', seed=1, **kw)
    if system == 3:
        kw.setdefault('task_info', {'stabilization_goal': [0.0, 0.0, 1.0], 'stabilization_goal_tolerance': 0.05})
    return scg.make('quadrotor', quad_type=system, num_envs=n, device='cuda', seed=1, **kw)


def _host_model(system, dt):
    import safe_control_gym_amd as scg
    kw = dict(num_envs=1, device='cpu', ctrl_freq=round(1 / dt), pyb_freq=round(1 / dt))
    if system == 'cartpole':
        return scg.make('cartpole', **kw).symbolic
    if system == 3:
        kw['task_info'] = {'stabilization_goal': [0.0, 0.0, 1.0], 'stabilization_goal_tolerance': 0.05}
    return scg.make('quadrotor', quad_type=system, **kw).symbolic


# ---- CPU: the NumPy mirror ---------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('system', SYSTEMS)
def test_host_fd_func_is_the_flow_to_the_stated_tolerance(system):
    dt = 0.02
    m = _host_model(system, dt)
    p = params_of(system)
    xs, us = sample(system, 6, np.random.default_rng(3))
    for x, u in zip(xs, us):
        exact = independent.exact_step(system, x, u, p, dt)
        got = m.fd_func(x0=x, p=u)['xf']
        assert np.all(np.abs(got - exact) <= CVODES_ATOL + CVODES_RTOL * np.abs(exact))
        tight = m.fd_func(x0=x, p=u, rtol=1e-11, atol=1e-13)['xf']
        assert np.all(np.abs(tight - exact) <= 1e-12 + 1e-10 * np.abs(exact)), np.abs(tight - exact).max()
        # one fixed RK4 step (rk_discrete) is the coarser answer to the same question
        rk = m.rk4_func(x0=x, p=u)['xf']
        assert np.abs(rk - exact).max() + 1e-13 >= np.abs(tight - exact).max()      # (1-D: both exact, the ODE is polynomial in t)
        assert np.allclose(rk, exact, rtol=1e-4, atol=1e-5)       # (up to ~1e-5 at pitch rates of several rad/s)


@pytest.mark.parametrize('system', SYSTEMS)
def test_host_fd_sensitivities_match_the_variational_equations(system):
    dt = 0.02
    m = _host_model(system, dt)
    p = params_of(system)
    xs, us = sample(system, 3, np.random.default_rng(4))
    out = m.fd_jac_func(x0=xs, p=us)                       # batched call
    for k, (x, u) in enumerate(zip(xs, us)):
        xf, sx, su = independent.exact_step_sensitivities(system, x, u, p, dt)
        assert np.all(np.abs(out['xf'][k] - xf) <= CVODES_ATOL + CVODES_RTOL * np.abs(xf))
        assert np.abs(out['jac_xf_x0'][k] - sx).max() <= 2e-7 * max(1.0, np.abs(sx).max())
        assert np.abs(out['jac_xf_p'][k] - su).max() <= 2e-7 * max(1.0, np.abs(su).max())


# ---- GPU: scg_model_flow -----------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize('system', SYSTEMS)
def test_device_flow_matches_dop853(gpu, system):
    import torch
    n = 150
    env = _make(system, n, ctrl_freq=50, pyb_freq=50)
    env.reset()
    before = env.state.clone()
    xs, us = sample(system, n, np.random.default_rng(31))
    xt = torch.as_tensor(xs, dtype=torch.float64, device='cuda')
    ut = torch.as_tensor(us, dtype=torch.float64, device='cuda')
    dt = env.CTRL_TIMESTEP
    p = params_of(system)
    xf, iv = env.model_flow(xt, ut, return_intervals=True)
    xf_tight, iv_tight = env.model_flow(xt, ut, rtol=1e-11, atol=1e-13, return_intervals=True)
    assert xf.dtype == torch.float64 and xf.shape == (n, xs.shape[1])
    assert torch.equal(env.state, before)                                    # the envs were not advanced
    assert (iv > 0).all() and (iv_tight >= iv).all()
    xf, xf_tight = xf.cpu().numpy(), xf_tight.cpu().numpy()
    worst = 0.0
    for k in range(0, n, 3):
        exact = independent.exact_step(system, xs[k], us[k], p, dt)
        assert np.all(np.abs(xf[k] - exact) <= CVODES_ATOL + CVODES_RTOL * np.abs(exact)), (k, xf[k], exact)
        worst = max(worst, np.max(np.abs(xf_tight[k] - exact) / (1e-2 + np.abs(exact))))
    assert worst <= 1e-10, worst
    # the host mirror runs the same scheme
    host = env.symbolic.fd_func(x0=xs[:8], p=us[:8])['xf']
    assert np.allclose(host, xf[:8], rtol=1e-12, atol=1e-13)
    # symbolic.fd_func on device tensors is this entry point, returned in the caller's dtype
    x32 = env.symbolic.fd_func(xt.float(), ut.float())['xf']
    assert x32.dtype == torch.float32 and torch.allclose(x32.double(), torch.as_tensor(xf, device='cuda'), rtol=2e-6, atol=2e-6)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('system', SYSTEMS)
def test_device_flow_sensitivities_match_the_variational_equations(gpu, system):
    import torch
    n = 70                                                   # not a multiple of 64 / (nx + nu): lanes of one env straddle waves
    env = _make(system, n, ctrl_freq=50, pyb_freq=50)
    env.reset()
    xs, us = sample(system, n, np.random.default_rng(32))
    xt = torch.as_tensor(xs, dtype=torch.float64, device='cuda')
    ut = torch.as_tensor(us, dtype=torch.float64, device='cuda')
    dt = env.CTRL_TIMESTEP
    p = params_of(system)
    xf, sx, su = env.model_flow(xt, ut, sensitivities=True)
    plain = env.model_flow(xt, ut)
    assert torch.equal(xf, plain)                            # the nominal of the sensitivity launch is the plain launch
    nx, nu = xs.shape[1], us.shape[1]
    assert sx.shape == (n, nx, nx) and su.shape == (n, nx, nu) and sx.dtype == torch.float64
    sx, su = sx.cpu().numpy(), su.cpu().numpy()
    worst = 0.0
    for k in range(0, n, 4):
        _, ex, eu = independent.exact_step_sensitivities(system, xs[k], us[k], p, dt)
        worst = max(worst, np.abs(sx[k] - ex).max() / max(1.0, np.abs(ex).max()), np.abs(su[k] - eu).max() / max(1.0, np.abs(eu).max()))
    assert worst <= 2e-7, worst                              # the nominal is accepted at 1e-6: its derivative inherits ~1e-8
    # at a tight tolerance the difference quotient itself is what is left
    _, sx_t, su_t = env.model_flow(xt, ut, rtol=1e-11, atol=1e-13, sensitivities=True)
    sx_t, su_t = sx_t.cpu().numpy(), su_t.cpu().numpy()
    worst = 0.0
    for k in range(0, n, 7):
        _, ex, eu = independent.exact_step_sensitivities(system, xs[k], us[k], p, dt)
        worst = max(worst, np.abs(sx_t[k] - ex).max() / max(1.0, np.abs(ex).max()), np.abs(su_t[k] - eu).max() / max(1.0, np.abs(eu).max()))
    assert worst <= 5e-9, worst
    jac = env.symbolic.fd_jac_func(xt, ut)
    assert np.array_equal(jac['jac_xf_x0'].cpu().numpy(), sx) and np.array_equal(jac['jac_xf_p'].cpu().numpy(), su)
    env.close()


@pytest.mark.gpu
def test_device_flow_step_control_and_failure_flag(gpu):
    """A long interval needs more RK4 steps; a budget that cannot meet the tolerance is reported, not hidden."""
    import torch
    n = 64
    env = _make(3, n)
    env.reset()
    xs, us = sample(3, n, np.random.default_rng(33))
    xs *= 0.4                                                # a tenth of a second must not reach the Euler-angle singularity
    xt = torch.as_tensor(xs, dtype=torch.float64, device='cuda')
    ut = torch.as_tensor(us, dtype=torch.float64, device='cuda')
    p = params_of(3)
    _, iv_short = env.model_flow(xt, ut, dt=1.0 / 240, rtol=1e-8, atol=1e-10, return_intervals=True)
    xf_long, iv_long = env.model_flow(xt, ut, dt=0.1, rtol=1e-8, atol=1e-10, return_intervals=True)
    assert (iv_long > 0).all() and iv_long.float().mean() > 2 * iv_short.float().mean(), (iv_short, iv_long)
    xf_long = xf_long.cpu().numpy()
    for k in range(0, n, 8):
        exact = independent.exact_step(3, xs[k], us[k], p, 0.1)
        assert np.all(np.abs(xf_long[k] - exact) <= 1e-10 + 1e-8 * np.abs(exact)), (k, xf_long[k] - exact)
    _, iv_fail = env.model_flow(xt, ut, dt=0.25, rtol=1e-13, atol=1e-15, max_doublings=0, return_intervals=True)
    assert (iv_fail == -2).all()
    bad = xt.clone()
    bad[5, 0] = float('nan')
    xf_bad, iv_bad = env.model_flow(bad, ut, return_intervals=True)
    assert iv_bad[5] < 0 and torch.isnan(xf_bad[5, 0]) and (iv_bad[:5] > 0).all()
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('system', [3, 'cartpole'])
def test_device_flow_uses_per_env_inertial_properties(gpu, system):
    import torch
    n = 40
    env = _make(system, n, randomized_inertial_prop=True, ctrl_freq=50, pyb_freq=50)
    env.reset()
    prop = env.inertial_properties.cpu().numpy().astype(np.float64)          # float32 planes, widened
    assert np.ptp(prop[:, 0]) > 0
    xs, us = sample(system, n, np.random.default_rng(34))
    xf = env.model_flow(torch.as_tensor(xs, device='cuda'), torch.as_tensor(us, device='cuda'), rtol=1e-10, atol=1e-12).cpu().numpy()
    for k in range(0, n, 5):
        exact = independent.exact_step(system, xs[k], us[k], params_of(system, list(prop[k])), env.CTRL_TIMESTEP)
        assert np.max(np.abs(xf[k] - exact) / (1e-2 + np.abs(exact))) <= 1e-9
    env.close()


@pytest.mark.gpu
def test_device_flow_semigroup_and_chain_rule_at_scale(gpu):
    """Size-independent properties at 2^18 quadrotor-3D states: flow(2 dt) = flow(dt) o flow(dt), and the sensitivities
    compose by the chain rule."""
    import torch
    n = 1 << 18
    env = _make(3, n, ctrl_freq=50, pyb_freq=50)
    env.reset()
    g = torch.Generator(device='cuda').manual_seed(5)
    xt = torch.randn((n, 12), generator=g, device='cuda', dtype=torch.float64) * 0.5
    xt[:, 6:9].clamp_(-1.0, 1.0)
    lo, hi = (float(np.min(v)) for v in ref_model.action_bounds(4))
    ut = torch.rand((n, 4), generator=g, device='cuda', dtype=torch.float64) * (hi - lo) + lo
    dt = env.CTRL_TIMESTEP
    kw = dict(rtol=1e-11, atol=1e-13)
    x1, s1, u1 = env.model_flow(xt, ut, dt=dt, sensitivities=True, **kw)
    x2, s2, u2 = env.model_flow(x1, ut, dt=dt, sensitivities=True, **kw)
    xd, sd, ud = env.model_flow(xt, ut, dt=2 * dt, sensitivities=True, **kw)
    scale = 1e-2 + xd.abs()
    assert ((x2 - xd).abs() / scale).max().item() <= 1e-9
    assert ((torch.bmm(s2, s1) - sd).abs().amax(dim=(1, 2)) / sd.abs().amax(dim=(1, 2)).clamp_min(1.0)).max().item() <= 2e-8
    assert (((torch.bmm(s2, u1) + u2) - ud).abs().amax(dim=(1, 2)) / ud.abs().amax(dim=(1, 2)).clamp_min(1.0)).max().item() <= 2e-8
    env.close()
