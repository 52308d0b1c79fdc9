#!/usr/bin/env python
"""Roofline line for the batched exact discretisation (scg_discretize_exact, csrc/scg_expm.hip).

Bound: the matrix cores.  Per 16 x 16 tile: kExpmDegree = 9 Horner products + s squarings, 4 v_mfma_f32_16x16x4_f32 each,
2 * 16^3 = 8192 flop per product.  f32-input MFMA peak = the f32 vector peak (157.3 TFLOP/s spec, MI355X_MICROARCH.md): the
same flop/clk/SIMD as VALU, one instruction per 1024 FMAs.  HBM traffic is 8 (nx + m) nx bytes per system (read A, B; write
Ad, Bd) -- reported beside it."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safe_control_gym_amd.controllers.mpc.mpc_utils import discretize_exact

out = []
for nx, m, scale, label in ((12, 4, 3.0, 'quadrotor-3D sized, ||A dt|| < 1/2 (no squaring)'),
                            (12, 4, 400.0, 'quadrotor-3D sized, ||A dt|| ~ 8 (5 squarings)'),
                            (6, 2, 3.0, 'quadrotor-2D sized (2 systems per wave)'),
                            (2, 1, 3.0, 'quadrotor-1D sized (4 systems per wave)')):
    N = 1 << 20
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    A = torch.randn((N, nx, nx), generator=g, device='cuda') * scale / nx
    B = torch.randn((N, nx, m), generator=g, device='cuda') * 30.0
    dt = 1 / 60
    for _ in range(3):
        Ad, Bd, _ = discretize_exact(A, B, dt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        discretize_exact(A, B, dt)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    norm = float((A.abs().sum(2).amax(1) * dt).mean())
    sq = max(0, int(torch.ceil(torch.log2(torch.tensor(norm / 0.5))).item()) + 0) if norm > 0.5 else 0
    per = 16 // (4 if nx + m <= 4 else 8 if nx + m <= 8 else 16)
    tiles = (N + per - 1) // per
    flops = tiles * (9 + sq) * 8192
    bytes_alg = N * 8 * (nx + m) * nx
    out.append({'kernel': 'scg_expm_kernel', 'case': label, 'N': N, 'nx': nx, 'm': m, 'mean_norm_inf_A_dt': norm, 'us': ms * 1e3,
                'systems_per_s': N / ms * 1e3, 'mfma_TFLOPs': flops / ms / 1e9, 'frac_of_f32_mfma_peak_157': flops / ms / 1e9 / 157.3,
                'hbm_GBs': bytes_alg / ms / 1e6})
    print(json.dumps(out[-1]))
