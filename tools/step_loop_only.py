"""A bare Python step loop (lean tier, 2^20 envs, 1500 steps) for tracing: tools/step_loop_trace.py reads the kernel trace."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import safe_control_gym_amd as scg
n = 1 << 20
env = scg.make('quadrotor', quad_type=3, num_envs=n, device='cuda', seed=1, randomized_init=True, task='traj_tracking', cost='quadratic', info_level='minimal')
env.reset()
lo, hi = float(env.action_space.low[0]), float(env.action_space.high[0])
seq = torch.rand((64, 4, env._ld), device='cuda') * (hi - lo) + lo
for k in range(1500):
    env.step(seq[k % 64])
torch.cuda.synchronize()
