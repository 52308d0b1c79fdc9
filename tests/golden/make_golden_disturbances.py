#!/usr/bin/env python
"""Generate tests/golden/disturbances.json by RUNNING the reference's disturbance classes
(/root/reference/safe_control_gym/envs/disturbances.py -- importable without casadi / pybullet / gym).

Each case records the constructor spec, the raw draws handed to the classes through a stub ``np_random``
(uniforms in [0, 1), or the per-episode ``randint``), the step counter, and the vector ``apply(target, env)``
returned.  tests/test_disturbances.py replays the cases through oracle/ref_disturbances.py.  Run in the build
container only (the reference never travels to the GPU box):  python tests/golden/make_golden_disturbances.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, '/root/reference')

from safe_control_gym.envs import disturbances as ref          # noqa: E402
from oracle.ref_disturbances import standard_normals              # noqa: E402  (Box-Muller of the raw uniforms)


class StubRandom:
    """np_random stand-in: hands the recorded raw draws to the reference code."""

    def __init__(self):
        self.u = None
        self.offset = None

    def uniform(self, low=0.0, high=1.0, size=None):
        n = int(np.prod(size)) if size is not None else 1
        return np.asarray(low) + (np.asarray(high) - np.asarray(low)) * self.u[:n]

    def normal(self, loc=0.0, scale=1.0, size=None):
        n = int(np.prod(size)) if size is not None else 1
        return loc + np.asarray(scale) * standard_normals(self.u)[:n]

    def randint(self, high):
        return self.offset


class StubEnv:
    EPISODE_LEN_SEC = 5
    CTRL_TIMESTEP = 1.0 / 60
    PYB_TIMESTEP = 1.0 / 240

    def __init__(self):
        self.np_random = StubRandom()
        self.ctrl_step_counter = 0
        self.pyb_step_counter = 0


SPECS = [
    (4, dict(disturbance_func='impulse', magnitude=2.5, step_offset=3, duration=1, decay_rate=1)),
    (4, dict(disturbance_func='impulse', magnitude=1.5, step_offset=2, duration=6, decay_rate=0.7, mask=[1, 0, 1, 0])),
    (2, dict(disturbance_func='impulse', magnitude=-0.4, step_offset=None, duration=5, decay_rate=0.5)),
    (3, dict(disturbance_func='step', magnitude=0.03, step_offset=4)),
    (3, dict(disturbance_func='step', magnitude=0.2, step_offset=None, mask=[0, 0, 1])),
    (4, dict(disturbance_func='uniform', low=-0.01, high=0.02)),
    (3, dict(disturbance_func='uniform', low=[-0.05, 0.0, 0.01], high=[0.05, 0.1, 0.02], mask=[1, 1, 0])),
    (6, dict(disturbance_func='white_noise', std=0.01)),
    (2, dict(disturbance_func='white_noise', std=[0.1, 0.3], mask=[0, 1])),
    (1, dict(disturbance_func='white_noise', std=2.0)),
    (4, dict(disturbance_func='periodic', scale=0.05, frequency=2.0)),
    (12, dict(disturbance_func='periodic', scale=1.0, frequency=0.5, mask=[0] * 12)),       # mask is ignored (:273)
]


def main():
    rng = np.random.default_rng(20240607)
    cases = []
    for dim, spec in SPECS:
        env = StubEnv()
        cfg = {k: v for k, v in spec.items() if k != 'disturbance_func'}
        obj = ref.DISTURBANCE_TYPES[spec['disturbance_func']](env, dim=dim, **cfg)
        obj.seed(env)
        offset = int(rng.integers(0, 12))
        env.np_random.offset = offset
        obj.reset(env)
        steps = []
        for t in range(0, 16):
            env.ctrl_step_counter = t
            env.pyb_step_counter = 4 * t
            u = (rng.integers(0, 1 << 24, size=(dim + 3) // 4 * 4) * 2.0 ** -24)
            env.np_random.u = u
            target = rng.normal(size=dim)
            out = np.asarray(obj.apply(target.copy(), env), dtype=np.float64)
            steps.append(dict(t=t, u=u.tolist(), target=target.tolist(), out=out.tolist()))
        cases.append(dict(dim=dim, spec=spec, episode_offset=offset, ctrl_timestep=env.CTRL_TIMESTEP, steps=steps))
    with open(os.path.join(HERE, 'disturbances.json'), 'w') as f:
        json.dump(dict(source='safe_control_gym/envs/disturbances.py run with a stub np_random', cases=cases), f, indent=0)
    print('wrote %d cases' % len(cases))


if __name__ == '__main__':
    main()
