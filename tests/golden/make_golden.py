#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ from the float64 oracle.

The reference itself cannot run here (casadi / pybullet / gym are not installed) and its
test-suite holds no golden vectors, so these fixtures are outputs of OUR restatement (``oracle/``) --
pinned by the known-answer tests and the independent derivation in tests/test_oracle_pinning.py.
They freeze the oracle: tests/test_oracle_pinning.py::test_golden_matches_oracle fails if the oracle
drifts, and the GPU parity test replays them through the HIP engine.

Each fixture: N envs, T free-running steps with auto-reset; inputs = per-env initial states, the
float32 action sequence, seed; outputs = per-step obs, reward, done, mse (+ constraint values).
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import zlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle.ref_env import RefEnv                      # noqa: E402
from oracle.ref_vec import RefVecEnv                   # noqa: E402
from helpers import PhiloxSampler                      # noqa: E402

COMPETITION_CONSTRAINTS = [                            # schema + numbers of competition/getting_started.yaml:123-156
    {'constraint_form': 'default_constraint', 'constrained_variable': 'input'},
    {'constraint_form': 'bounded_constraint', 'constrained_variable': 'state', 'active_dims': [0, 2, 4],
     'upper_bounds': [3, 3, 2], 'lower_bounds': [-3, -3, -0.1]},
]
VERBOSE_API_CONSTRAINTS = [                            # tests/scripts/verbose_api.yaml:151-155
    {'constraint_form': 'default_constraint', 'constrained_variable': 'input'},
    {'constraint_form': 'default_constraint', 'constrained_variable': 'state'},
]

CASES = {
    'quad3d_track_quadratic': dict(system=3, kw=dict(task='traj_tracking', cost='quadratic', ctrl_freq=60, pyb_freq=240)),
    'quad3d_stab_rl_constraints': dict(system=3, kw=dict(
        task='stabilization', cost='rl_reward', ctrl_freq=60, pyb_freq=240,
        task_info={'stabilization_goal': [0.1, -0.1, 1.0], 'stabilization_goal_tolerance': 0.05},
        constraints=COMPETITION_CONSTRAINTS)),
    'quad2d_track_quadratic_constraints': dict(system=2, kw=dict(
        task='traj_tracking', cost='quadratic', ctrl_freq=60, pyb_freq=240, episode_len_sec=10,
        constraints=VERBOSE_API_CONSTRAINTS, done_on_violation=True)),
    'quad1d_stab_rl': dict(system=1, kw=dict(task='stabilization', cost='rl_reward', ctrl_freq=50, pyb_freq=50)),
    'cartpole_stab_rl': dict(system='cartpole', kw=dict(task='stabilization', cost='rl_reward', ctrl_freq=50, pyb_freq=50)),
}
N, T, SEED = 8, 40, 1337


def run_case(name, system, kw):
    sampler = PhiloxSampler(SEED, system)
    refs = [RefEnv(system=system, env_index=i, reset_sampler=sampler, randomized_init=True, **kw) for i in range(N)]
    vec = RefVecEnv(refs)
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    lo, hi = refs[0].action_space_low, refs[0].action_space_high
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    hover = refs[0].U_GOAL
    # mostly near the equilibrium input, a few lanes with wide random actions (they go out of bounds and reset)
    spread = np.where(np.arange(N) < 6, 0.08, 1.1)[:, None]
    actions = np.empty((T, N, refs[0].nu), dtype=np.float32)
    for k in range(T):
        centre = np.where(np.arange(N)[:, None] < 6, hover[None, :], mid[None, :])
        actions[k] = (centre + spread * half[None, :] * rng.uniform(-1, 1, size=(N, refs[0].nu))).astype(np.float32)
    sampler.serial = 0
    obs0, _ = vec.reset()
    out = dict(obs0=obs0, actions=actions)
    obs, rew, done, mse, cval, viol = [], [], [], [], [], []
    for k in range(T):
        sampler.serial = k + 1
        o, r, d, info = vec.step(actions[k].astype(np.float64))
        term = [inf.get('terminal_info', inf) for inf in info['n']]
        obs.append(o); rew.append(r); done.append(d)
        mse.append([t['mse'] for t in term])
        if refs[0].constraints is not None:
            cval.append(np.stack([t['constraint_values'] for t in term]))
            viol.append([t['constraint_violation'] for t in term])
    out.update(obs=np.stack(obs), reward=np.stack(rew), done=np.stack(done).astype(np.uint8), mse=np.array(mse))
    if cval:
        out.update(constraint_values=np.stack(cval), constraint_violation=np.array(viol, dtype=np.uint8))
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    meta = {'seed': SEED, 'n': N, 't': T, 'cases': {}}
    for name, case in CASES.items():
        data = run_case(name, case['system'], case['kw'])
        np.savez_compressed(os.path.join(here, name + '.npz'), **data)
        meta['cases'][name] = {'system': case['system'], 'kw': case['kw'], 'dones': int(data['done'].sum())}
        print(name, {k: v.shape for k, v in data.items()}, 'dones', int(data['done'].sum()))
    with open(os.path.join(here, 'cases.json'), 'w') as f:
        json.dump(meta, f, indent=1)


if __name__ == '__main__':
    main()
