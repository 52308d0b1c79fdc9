#!/usr/bin/env python
"""Roofline line for the observation normaliser (next row 8f-4): [d][N] planes, two passes.
Algorithmic bytes per element: 4 (moments read) + 4 (apply read) + 4 (write) = 12 B; the second read can be
served by the Infinity Cache, so HBM traffic may be lower than that."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safe_control_gym_amd.math_and_models.normalization import MeanStdNormalizer, RewardStdNormalizer

d = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
ld = N + 17 * 64
planes = torch.randn((d, ld), device='cuda')
obs = planes[:, :N].T
norm = MeanStdNormalizer(shape=(d,))
for _ in range(5):
    norm(obs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    norm(obs)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
b = 12 * d * N
out = {'kernel': 'scg_mean_std_normalize', 'd': d, 'N': N, 'us': ms * 1e3, 'achieved_GBs': b / ms / 1e6,
       'frac_of_8TBs': b / ms / 1e6 / 8000}
rew, done = torch.randn(N, device='cuda'), torch.rand(N, device='cuda') < 0.01
rn = RewardStdNormalizer()
for _ in range(5):
    rn(rew, done)
torch.cuda.synchronize()
e0.record()
for _ in range(reps):
    rn(rew, done)
e1.record(); torch.cuda.synchronize()
out['reward_us'] = e0.elapsed_time(e1) / reps * 1e3
print(json.dumps(out))
