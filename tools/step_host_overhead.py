#!/usr/bin/env python
"""Host cost of one Python env.step() call: at small batches the loop is bound by it, not by the kernel.
Prints per-call wall time of a tight step loop (no sync inside) for several configurations and a cProfile of one of them.

    python tools/step_host_overhead.py [num_envs] [steps]
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import safe_control_gym_amd as scg


def loop(env, act, steps):
    for _ in range(steps):
        env.step(act)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    base = dict(quad_type=3, num_envs=n, device='cuda', seed=1, randomized_init=True, task='traj_tracking', cost='quadratic')
    cases = [('minimal_views', dict(info_level='minimal')), ('minimal_fresh', dict(info_level='minimal', copy_outputs=True)),
             ('full_views', dict(copy_outputs=False)), ('full_fresh (default make())', dict())]
    for name, kw in cases:
        env = scg.make('quadrotor', **base, **kw)
        env.reset()
        lo, hi = float(env.action_space.low[0]), float(env.action_space.high[0])
        act = torch.rand((n, 4), device='cuda') * (hi - lo) + lo
        loop(env, act, 200)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(env, act, steps)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print('%-30s n=%d: host %.2f us per step() call, with the device drained %.2f us/step' %
              (name, n, t_host / steps * 1e6, t_all / steps * 1e6), flush=True)
        if name.startswith('full_fresh'):
            pr = cProfile.Profile()
            pr.enable()
            loop(env, act, steps)
            pr.disable()
            torch.cuda.synchronize()
            pstats.Stats(pr).sort_stats('tottime').print_stats(14)
        env.close()


if __name__ == '__main__':
    main()
