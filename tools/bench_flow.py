#!/usr/bin/env python
"""Throughput of scg_model_flow (fd_func as the reference defines it: the error-controlled flow over [0, dt], float64) and of its
forward sensitivities.  Compute-bound on the fp64 VALU: the figure of merit is RK4 steps (4 evaluations of f) per second.
usage: bench_flow.py [system=3] [log2 n=20] [log2 n_sens=16]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_control_gym_amd as scg

system = sys.argv[1] if len(sys.argv) > 1 else '3'
system = int(system) if system != 'cartpole' else system
N = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
NS = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 16)


def make(n):
    if system == 'cartpole':
        return scg.make('cartpole', num_envs=n, device='cuda', seed=1)
    kw = {'task_info': {'stabilization_goal': [0.0, 0.0, 1.0], 'stabilization_goal_tolerance': 0.05}} if system == 3 else {}
    return scg.make('quadrotor', quad_type=system, num_envs=n, device='cuda', seed=1, **kw)


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


out = {'kernel': 'scg_model_flow', 'system': system}
for tag, n, sens in (('flow', N, False), ('flow_with_sensitivities', NS, True)):
    env = make(n)
    env.reset()
    eng = env._engine
    nx, nu = env.state_dim, env.action_dim
    g = torch.Generator(device='cuda').manual_seed(3)
    xs = (torch.randn((nx, n), generator=g, device='cuda', dtype=torch.float64) * 0.4).clamp_(-1.0, 1.0).contiguous()   # (attitude away from the Euler-rate singularity: one lane there takes thousands of steps)
    us = torch.full((nu, n), float(env.U_EQ[0]), device='cuda', dtype=torch.float64)
    for name, rtol, atol in (('cvodes_default_tolerances', 1e-6, 1e-8), ('tight_1e-11', 1e-11, 1e-13)):
        ms = timed(lambda: eng.model_flow(xs, us, None, n, n, env.CTRL_TIMESTEP, rtol, atol, want_sens=sens))
        iv = eng.model_flow(xs, us, None, n, n, env.CTRL_TIMESTEP, rtol, atol, want_sens=sens)[3].double()
        m2 = iv.abs().mean().item()                       # 2m of the accepted pair
        # RK4 steps evaluated: the nominal runs 1 + 2 + 4 + ... + 2m = 4m - 1 steps; each sensitivity lane repeats it and adds 2 (m + 2m)
        steps = (2 * m2 - 1) * (n * ((nx + nu) if sens else 1)) + (3 * m2 * n * (nx + nu) if sens else 0)
        out['%s/%s' % (tag, name)] = {'n': n, 'ms': ms, 'flows_per_s': n / ms * 1e3, 'mean_accepted_steps': m2,
                                      'rk4_steps_per_s': steps / ms * 1e3}
    env.close()
print(json.dumps(out))
