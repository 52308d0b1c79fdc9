#!/usr/bin/env python
"""Roofline line for the returns/GAE kernel (next row 8f-1): T x N planes, 24 algorithmic bytes per (t, env)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safe_control_gym_amd.controllers.ppo.ppo_utils import compute_returns_and_advantages

T, N = 128, 1 << 20
g = torch.Generator(device='cuda'); g.manual_seed(0)
rews, vals = torch.randn(T, N, 1, generator=g, device='cuda'), torch.randn(T, N, 1, generator=g, device='cuda')
masks = (torch.rand(T, N, 1, generator=g, device='cuda') > 0.01).float()
lv = torch.randn(N, 1, generator=g, device='cuda')
for _ in range(3):
    compute_returns_and_advantages(rews, vals, masks, 0, lv, use_gae=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    compute_returns_and_advantages(rews, vals, masks, 0, lv, use_gae=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
bytes_alg = 20 * T * N          # rew, val, mask reads + ret, adv writes (no terminal_v plane in this run)
print(json.dumps({'kernel': 'scg_gae', 'T': T, 'N': N, 'ms': ms, 'achieved_GBs': bytes_alg / ms / 1e6,
                  'frac_of_8TBs': bytes_alg / ms / 1e6 / 8000, 'transitions_per_s': T * N / ms * 1e3}))
