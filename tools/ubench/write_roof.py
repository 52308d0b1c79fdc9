#!/usr/bin/env python
"""What a pure write stream / a 1:1 copy / a read-only reduction reach on this box (torch's own elementwise kernels, 16-byte accesses):
the ceilings the write-dominated next-row kernels (scg_linearize: 92 % writes) are judged against."""
import json
import torch

N = 200 * (1 << 20)          # 200 Mi floats = 800 MiB
a = torch.empty(N, device='cuda')
b = torch.empty(N, device='cuda')


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {}
ms = timed(lambda: a.fill_(1.5)); out['fill_800MiB'] = {'ms': ms, 'GBs': 4 * N / ms / 1e6}
ms = timed(lambda: b.copy_(a)); out['copy_800MiB_to_800MiB'] = {'ms': ms, 'GBs': 8 * N / ms / 1e6}
ms = timed(lambda: a.sum()); out['sum_800MiB'] = {'ms': ms, 'GBs': 4 * N / ms / 1e6}
ms = timed(lambda: torch.add(a, 1.0, out=b)); out['add_scalar_read_write'] = {'ms': ms, 'GBs': 8 * N / ms / 1e6}
print(json.dumps(out))
