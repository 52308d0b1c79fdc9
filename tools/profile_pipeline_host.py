#!/usr/bin/env python
"""Where the HOST time of an eager PPO-style rollout goes: cProfile over examples/ppo_rollout.py-like steps at a batch small
enough that the device is never the limit.  python tools/profile_pipeline_host.py [envs] [steps]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import safe_control_gym_amd as scg
from safe_control_gym_amd.controllers.ppo.ppo_utils import PPOBuffer
from safe_control_gym_amd.math_and_models.normalization import MeanStdNormalizer, RewardStdNormalizer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
env = scg.make('quadrotor', quad_type=3, num_envs=n, device='cuda', seed=0, task='traj_tracking', cost='rl_reward',
               normalized_rl_action_space=True, obs_goal_horizon=1, randomized_init=True, copy_outputs=False)
od, nu = env.obs_dim, env.action_dim
obs_norm, rew_norm = MeanStdNormalizer(shape=(od,), clip=10.0), RewardStdNormalizer(gamma=0.99, clip=10.0)
buf = PPOBuffer(env.observation_space, env.action_space, T, n, env=env)
act = torch.zeros((nu, env.plane_stride), device='cuda')
obs_norm(env.reset(), out=buf.obs_row(0))


def rollout(parts):
    for t in range(T):
        if 'step' in parts:
            raw, rew, done, info = env.step(act, out=buf.step_out(t, obs=False, reward=False))
        if 'obs_norm' in parts:
            obs_norm(raw, out=buf.obs_row(t + 1))
        if 'rew_norm' in parts:
            rew_norm(rew, done, out=buf.rew_row(t))


for parts in (('step',), ('step', 'obs_norm'), ('step', 'obs_norm', 'rew_norm')):
    rollout(parts); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        rollout(parts)
    torch.cuda.synchronize()
    print('%-28s %.2f us per step (host clock, %d envs)' % ('+'.join(parts), (time.perf_counter() - t0) / 5 / T * 1e6, n))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    rollout(('step', 'obs_norm', 'rew_norm'))
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
env.close()
