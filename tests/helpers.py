"""Shared test plumbing: build an (engine env, oracle vector env) pair from one config and compare them.

The oracle (``oracle/``) is the checker; the engine (``safe_control_gym_amd``) is the thing checked.
"""
import numpy as np

from oracle import philox
from oracle.ref_env import RefEnv, INIT_STATE_LABELS, QUAD_INIT_STATE_RAND_INFO, CARTPOLE_INIT_STATE_RAND_INFO
from oracle.ref_vec import RefVecEnv

RTOL = 1e-5          # BASELINE.json north_star: 1e-5 relative, fp32


def _where(a, b):
    """Describe the worst element: index, engine value, oracle value."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    r = np.abs(a - b) / np.maximum(np.abs(b), 1.0)
    k = np.unravel_index(np.argmax(r), r.shape)
    return 'worst at %s: engine %.9g oracle %.9g' % (k, a[k], b[k])


def rel_err(a, b):
    """max |a - b| / max(|b|, 1): relative for large values, absolute near zero."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)))


class PhiloxSampler:
    """reset_sampler for RefEnv reproducing the device kernel's reset draws."""

    def __init__(self, seed, system, rand_info=None, env_index_base=0):
        labels = INIT_STATE_LABELS[system]
        info = rand_info
        if info is None:
            info = CARTPOLE_INIT_STATE_RAND_INFO if system == 'cartpole' else QUAD_INIT_STATE_RAND_INFO
        from safe_control_gym_amd.envs.randomization import parse_distribution
        parsed = [parse_distribution(info[k], k) if k in info else (0, 0.0, 0.0, (0.0, 0.0, 0.0)) for k in labels]
        self.kinds = np.array([q[0] for q in parsed])
        self.low = np.array([q[1] for q in parsed])
        self.high = np.array([q[2] for q in parsed])
        self.params = np.array([q[3] for q in parsed], dtype=np.float64)
        self.seed, self.base, self.serial = seed, env_index_base, 0

    def __call__(self, env_index, episode_index):
        return philox.reset_offsets(self.seed, self.base + env_index, self.serial, self.low, self.high, self.kinds, self.params)


def make_pair(system, n, seed=1337, device='cuda', engine_kwargs=None, init_states=None, **kw):
    """-> (engine_env, oracle RefVecEnv, sampler).  ``kw`` are the reference's env kwargs."""
    import safe_control_gym_amd as scg
    ek = dict(kw)
    ek.update(engine_kwargs or {})
    if system == 'cartpole':
        env = scg.make('cartpole', num_envs=n, device=device, seed=seed, **ek)
    else:
        env = scg.make('quadrotor', quad_type=system, num_envs=n, device=device, seed=seed, **ek)
    sampler = PhiloxSampler(seed, system, kw.get('init_state_randomization_info'))
    okw = {k: v for k, v in kw.items() if k not in ('physics', 'disturbances')}
    if 'physics' in kw:
        okw['integrator'] = 'euler' if kw['physics'] in ('euler', 'dyn') else 'rk4'
        okw['ground_effect'] = kw['physics'] in ('pyb_gnd', 'pyb_gnd_drag_dw')
        okw['drag'] = kw['physics'] in ('pyb_drag', 'pyb_gnd_drag_dw')
    for k_engine, k_oracle in (('integration_substeps', 'substeps'),):
        if k_engine in okw:
            okw[k_oracle] = okw.pop(k_engine)
    refs = []
    for i in range(n):
        o = dict(okw)
        if init_states is not None:
            o['init_state'] = np.asarray(init_states[i], dtype=np.float64)
        refs.append(RefEnv(system=system, env_index=i, reset_sampler=sampler, **o))
    dist = (engine_kwargs or {}).get('disturbances') or kw.get('disturbances')
    if dist:
        from oracle.ref_disturbances import RefDisturbanceModel
        for i, r in enumerate(refs):
            r.disturbance_model = RefDisturbanceModel(dist, system, seed, i, r.EPISODE_LEN_SEC, r.CTRL_TIMESTEP)
    if init_states is not None:
        env.set_init_states(np.asarray(init_states, dtype=np.float32))
    return env, RefVecEnv(refs), sampler


def _sync_serial(env, ref, sampler):
    sampler.serial = env._serial
    for r in ref.envs:
        if r.disturbance_model is not None:
            r.disturbance_model.serial = env._serial


def sync_reset(env, ref, sampler):
    _sync_serial(env, ref, sampler)
    obs_e = env.reset()
    if isinstance(obs_e, tuple):
        obs_e = obs_e[0]
    obs_r, _ = ref.reset()
    return obs_e, obs_r


def sync_step(env, ref, sampler, actions):
    """Step both with the same (N, nu) float32 actions."""
    _sync_serial(env, ref, sampler)
    a32 = np.asarray(actions, dtype=np.float32)
    out_e = env.step(a32)
    out_r = ref.step(a32.astype(np.float64))
    return out_e, out_r


def force_state(env, ref):
    """Teacher forcing: overwrite the engine state with the oracle's (rounded to fp32)."""
    import torch
    x = np.stack([r.state for r in ref.envs]).astype(np.float32)
    env.state_soa.copy_(torch.as_tensor(x.T, device=env.device))
    t = np.array([r.ctrl_step_counter for r in ref.envs], dtype=np.int32)
    env.ctrl_step_counter.copy_(torch.as_tensor(t, device=env.device))
    if 'ep_return' in env._bufs:
        env.episode_return.copy_(torch.as_tensor(ref.episode_return.astype(np.float32), device=env.device))
        env.episode_length.copy_(torch.as_tensor(ref.episode_length.astype(np.int32), device=env.device))
    for r, xi in zip(ref.envs, x):
        r.state = xi.astype(np.float64)


def ill_conditioned_lanes(ref, info_r, theta_guard=1.5):
    """Lanes whose post-step state lies past the Euler-angle singularity guard.

    The prior model's Euler-rate matrix has tan(theta) and 1/cos(theta) (quadrotor.py:671-673); the
    reference guards it by terminating at |theta| > 85 deg (quadrotor.py:745-746,979-993).  A terminal
    step that crosses |theta| = 90 deg amplifies a 1-ulp input perturbation by 1e4 in the float64 oracle
    itself (measured: DESIGN.md section 5), so values of such lanes are not comparable at any precision;
    they must still agree on `done`.  Only the 3-D quadrotor has this singularity.
    """
    n = len(ref.envs)
    bad = np.zeros(n, dtype=bool)
    if ref.envs[0].system != 3:
        return bad
    for i, inf in enumerate(info_r['n']):
        if 'terminal_observation' in inf:
            x = inf['terminal_observation']
            # ... or a stage of the step crossed it: roll / yaw thrown far outside their +-85 / +-180 degree boxes within ONE
            # step (from inside the box that takes > 100 rad/s of Euler rate, which only tan(theta), 1/cos(theta) deliver)
            bad[i] = (not np.all(np.isfinite(x))) or abs(x[7]) >= theta_guard or abs(x[6]) > np.pi or abs(x[8]) > 2 * np.pi
        elif theta_guard < 1.5:
            bad[i] = abs(ref.envs[i].state[7]) >= theta_guard
    return bad


def compare_step(out_e, out_r, ref, rtol=RTOL, check_info=True, theta_guard=1.5):
    """Assert the four return values (and info) of one step agree; returns the worst relative error."""
    obs_e, rew_e, done_e, info_e = out_e
    obs_r, rew_r, done_r, info_r = out_r
    worst = 0.0
    d_e = done_e.cpu().numpy().astype(bool)
    assert np.array_equal(d_e, done_r.astype(bool)), 'done mismatch at envs %s' % np.nonzero(d_e != done_r)[0][:8]
    bad = ill_conditioned_lanes(ref, info_r, theta_guard)
    if theta_guard >= 1.5:
        assert not np.any(bad & ~d_e), 'an ill-conditioned lane is not terminal'
    ok = ~bad
    okobs = ok | d_e            # a done lane returns its (well-conditioned) reset observation
    e = rel_err(obs_e.cpu().numpy()[okobs], obs_r[okobs]); worst = max(worst, e)
    assert e <= rtol, 'obs rel err %.3e; %s' % (e, _where(obs_e.cpu().numpy()[okobs], obs_r[okobs]))
    e = rel_err(rew_e.cpu().numpy()[ok], rew_r[ok]); worst = max(worst, e)
    assert e <= rtol, 'reward rel err %.3e; %s' % (e, _where(rew_e.cpu().numpy()[ok], rew_r[ok]))
    if not check_info:
        return worst
    infos = info_r['n']
    term = [inf.get('terminal_info', inf) for inf in infos]
    if 'mse' in info_e:
        e = rel_err(info_e['mse'].cpu().numpy()[ok], np.array([t['mse'] for t in term])[ok]); worst = max(worst, e)
        assert e <= rtol, 'mse rel err %.3e' % e
    if 'constraint_values' in info_e:
        cv = np.stack([t['constraint_values'] for t in term])
        e = rel_err(info_e['constraint_values'].cpu().numpy()[ok], cv[ok]); worst = max(worst, e)
        assert e <= rtol, 'constraint_values rel err %.3e; %s' % (e, _where(info_e['constraint_values'].cpu().numpy()[ok], cv[ok]))
        viol = np.array([t['constraint_violation'] for t in term])
        assert np.array_equal(info_e['constraint_violation'].cpu().numpy().astype(int)[ok], viol[ok]), 'violation mismatch'
    if 'TimeLimit.truncated' in info_e:
        tr = np.array([bool(t.get('TimeLimit.truncated', False)) for t in term])
        assert np.array_equal(info_e['TimeLimit.truncated'].cpu().numpy().astype(bool), tr), 'truncated mismatch'
    if 'goal_reached' in info_e:
        gr = np.array([bool(t.get('goal_reached', False)) for t in term])
        assert np.array_equal(info_e['goal_reached'].cpu().numpy().astype(bool), gr), 'goal_reached mismatch'
    idx = np.nonzero(d_e & ok)[0]
    if idx.size:
        if 'terminal_observation' in info_e:
            to = np.stack([infos[i]['terminal_observation'] for i in idx])
            te = info_e['terminal_observation'].cpu().numpy()[idx][:, :to.shape[1]]
            e = rel_err(te, to); worst = max(worst, e)
            assert e <= rtol, 'terminal_observation rel err %.3e; %s' % (e, _where(te, to))
        if 'episode' in info_e:
            er = np.array([infos[i]['episode']['r'] for i in idx])
            el = np.array([infos[i]['episode']['l'] for i in idx])
            e = rel_err(info_e['episode']['r'].cpu().numpy()[idx], er); worst = max(worst, e)
            assert e <= 10 * rtol, 'episode return rel err %.3e' % e      # a sum of up to CTRL_STEPS fp32 rewards
            assert np.array_equal(info_e['episode']['l'].cpu().numpy()[idx], el.astype(int)), 'episode length mismatch'
    return worst
