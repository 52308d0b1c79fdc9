"""A bare Python step loop (2^20 envs, dependent launches on one queue) for tracing: tools/step_loop_trace.py reads the kernel trace.
usage: step_loop_only.py [steps] [lean|default] [ring K]   (positional, so that the round-2 / round-3 trees run it unchanged)"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import safe_control_gym_amd as scg
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
tier = sys.argv[2] if len(sys.argv) > 2 else 'lean'
ring = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = 1 << 20
kw = dict(info_level='minimal') if tier == 'lean' else {}
if ring:
    kw['output_ring'] = ring
env = scg.make('quadrotor', quad_type=3, num_envs=n, device='cuda', seed=1, randomized_init=True, task='traj_tracking', cost='quadratic', **kw)
env.reset()
lo, hi = float(env.action_space.low[0]), float(env.action_space.high[0])
seq = torch.rand((64, 4, env._ld), device='cuda') * (hi - lo) + lo
for k in range(steps):
    env.step(seq[k % 64])
torch.cuda.synchronize()
