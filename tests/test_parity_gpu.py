"""GPU parity: the fused HIP step (through the C ABI) against the float64 oracle, same inputs.

Tolerance: 1e-5 relative (BASELINE.json north_star), measured as |a-b| / max(|b|, 1).
"""
import numpy as np
import pytest

from helpers import make_pair, sync_reset, sync_step, force_state, compare_step, rel_err, RTOL

pytestmark = pytest.mark.gpu


def _actions(env, rng, n, spread=1.0):
    lo, hi = env.action_space.low.astype(np.float64), env.action_space.high.astype(np.float64)
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    return (mid + spread * half * rng.uniform(-1.2, 1.2, size=(n, lo.shape[0]))).astype(np.float32)


CONFIGS = [
    # (id, system, kwargs)
    ('q3_track_quadratic', 3, dict(task='traj_tracking', cost='quadratic', ctrl_freq=60, pyb_freq=240)),
    ('q3_stab_rl', 3, dict(task='stabilization', cost='rl_reward', ctrl_freq=60, pyb_freq=240,
                           task_info={'stabilization_goal': [0.2, -0.1, 1.0], 'stabilization_goal_tolerance': 0.05})),
    ('q2_stab_rl', 2, dict(task='stabilization', cost='rl_reward', ctrl_freq=60, pyb_freq=240)),
    ('q2_track_rl', 2, dict(task='traj_tracking', cost='rl_reward', ctrl_freq=50, pyb_freq=50)),
    ('q1_stab_quadratic', 1, dict(task='stabilization', cost='quadratic', ctrl_freq=50, pyb_freq=50)),
    ('q1_track_rl', 1, dict(task='traj_tracking', cost='rl_reward', ctrl_freq=60, pyb_freq=240)),
    ('cartpole_stab_rl', 'cartpole', dict(task='stabilization', cost='rl_reward', ctrl_freq=50, pyb_freq=50)),
    ('cartpole_track_quadratic', 'cartpole', dict(task='traj_tracking', cost='quadratic', ctrl_freq=50, pyb_freq=50)),
]


@pytest.mark.parametrize('name,system,kw', CONFIGS, ids=[c[0] for c in CONFIGS])
def test_single_step_parity_teacher_forced(gpu, name, system, kw):
    """Identical state and action every step (the oracle's state is forced into the engine):
    next state, reward, done, mse, episode stats within 1e-5 for 40 steps incl. auto-resets."""
    n = 96                       # not a multiple of 64: exercises the ragged tail of the last wave
    env, ref, sampler = make_pair(system, n, randomized_init=True, **kw)
    obs_e, obs_r = sync_reset(env, ref, sampler)
    assert rel_err(obs_e.cpu().numpy(), obs_r) <= RTOL
    rng = np.random.default_rng(7)
    worst = 0.0
    for k in range(40):
        force_state(env, ref)
        out = sync_step(env, ref, sampler, _actions(env, rng, n))
        worst = max(worst, compare_step(*out, ref))
    print('%s worst rel err %.3e' % (name, worst))
