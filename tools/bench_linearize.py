#!/usr/bin/env python
"""Roofline line for the batched-Jacobian kernel (next row 8f-2).  Algorithmic bytes per env: 4 (nx + nu) read,
4 nx (nx + nu) written -- 3-D: 64 + 768 = 832 B.  SCG_LIN_DIRECT=1 times the un-staged store variant."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_control_gym_amd as scg
from safe_control_gym_amd import _capi

system = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
env = scg.make('quadrotor', quad_type=system, num_envs=N, device='cuda', seed=1, randomized_init=True,
               **({'task_info': {'stabilization_goal': [0.0, 0.0, 1.0], 'stabilization_goal_tolerance': 0.05}} if system == 3 else {}))
env.reset()
nx, nu = env.state_dim, env.action_dim
us = torch.full((nu, env._ld), float(env.U_EQ[0]), device='cuda')
A = torch.empty((N, nx, nx), device='cuda')
B = torch.empty((N, nx, nu), device='cuda')
lib, h, p = _capi.load(), env._engine.handle, _capi.ptr
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
call = lambda: lib.scg_linearize(h, p(env._bufs['state']), p(us), None, p(A), p(B), 0.0, env._ld, N, stream)
for _ in range(5):
    assert call() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    call()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
bytes_alg = N * 4 * (nx + nu) * (1 + nx)
print(json.dumps({'kernel': 'scg_linearize', 'system': system, 'N': N, 'direct': os.environ.get('SCG_LIN_DIRECT', '0'),
                  'us': ms * 1e3, 'achieved_GBs': bytes_alg / ms / 1e6, 'frac_of_8TBs': bytes_alg / ms / 1e6 / 8000,
                  'linearisations_per_s': N / ms * 1e3}))
