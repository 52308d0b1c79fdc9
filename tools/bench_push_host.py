#!/usr/bin/env python
"""Host time of PPOBuffer.push (the reference's per-step call, controllers/ppo/ppo.py:276-277) with device tensors at a batch where
the device is never the limit: cached row views + no conversion of what already is a float32 device tensor, against the former
form (index the (T, N, ...) tensor, as_tensor, reshape per key).  python tools/bench_push_host.py [envs]"""
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from safe_control_gym_amd.controllers.ppo.ppo_utils import PPOBuffer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = 128
buf = PPOBuffer(SimpleNamespace(shape=(12,)), SimpleNamespace(shape=(4,)), T, n)
batch = {'obs': torch.randn(n, 12, device='cuda'), 'act': torch.randn(n, 4, device='cuda'), 'rew': torch.randn(n, device='cuda'),
         'mask': torch.ones(n, 1, device='cuda'), 'v': torch.randn(n, 1, device='cuda'), 'logp': torch.randn(n, 1, device='cuda'),
         'terminal_v': torch.zeros(n, 1, device='cuda')}


def former_push(self, batch):
    for k, v in batch.items():
        shape = self.scheme[k]['vshape'][1:]
        self.__dict__['_mask' if k == 'mask' else k][self.t].copy_(torch.as_tensor(v, dtype=torch.float32, device=self.device).reshape(shape))
    self.t = (self.t + 1) % self.max_length


def timed(fn, calls=3000):
    for _ in range(300):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / calls * 1e6


for rnd in range(3):
    print('round %d  push of 7 keys, %d envs: one launch (scg_copy_segments) %.2f us, former form (one copy_ per key) %.2f us (host clock)'
          % (rnd, n, timed(lambda: buf.push(batch)), timed(lambda: former_push(buf, batch))), flush=True)
