#!/usr/bin/env python
"""Host time of a normaliser call that returns a FRESH tensor (no out=) at a batch where the device is never the limit: the
recycled result buffers (utils/recycle.py) against one torch.empty + views per call.  python tools/bench_normalizer_host.py [envs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from safe_control_gym_amd.math_and_models import normalization as nz

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = 12
planes = torch.randn((d, n), device='cuda')
x = planes.T                                   # an [n, d] view of SoA planes: what env.step() returns
r = torch.randn(n, device='cuda')
done = torch.zeros(n, dtype=torch.bool, device='cuda')


def timed(fn, calls=4000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / calls * 1e6


take = nz._FreshOutputs.take
for rnd in range(3):
    for label, pooled in (('recycled', True), ('allocating', False)):
        nz._FreshOutputs.take = take if pooled else (lambda self, *a: None)
        obs_norm, rew_norm = nz.MeanStdNormalizer(shape=(d,)), nz.RewardStdNormalizer(gamma=0.99)
        print('round %d %-10s obs normaliser %.2f us per call, reward normaliser %.2f us per call (%d envs, host clock)'
              % (rnd, label, timed(lambda: obs_norm(x)), timed(lambda: rew_norm(r, done)), n), flush=True)
nz._FreshOutputs.take = take
