#!/usr/bin/env python
"""The rollout half of PPO (reference: controllers/ppo/ppo.py:247-293) with everything on the device:
env.step -> observation / reward normalisers -> time-limit bootstrap -> PPOBuffer -> returns + GAE.

The policy here is a small random MLP (no learning): the point is the data path.  Nothing leaves the GPU
between the policy's action and the advantages.

    python examples/ppo_rollout.py [num_envs] [rollout_steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import safe_control_gym_amd as scg
from safe_control_gym_amd.controllers.ppo.ppo_utils import PPOBuffer, compute_returns_and_advantages
from safe_control_gym_amd.math_and_models.normalization import MeanStdNormalizer, RewardStdNormalizer


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    env = scg.make('quadrotor', quad_type=2, num_envs=n, device='cuda', seed=0, task='traj_tracking', cost='rl_reward',
                   normalized_rl_action_space=True, obs_goal_horizon=1, randomized_init=True)
    od, nu = env.obs_dim, env.action_dim
    torch.manual_seed(0)
    actor = torch.nn.Sequential(torch.nn.Linear(od, 64), torch.nn.Tanh(), torch.nn.Linear(64, nu), torch.nn.Tanh()).cuda()
    critic = torch.nn.Sequential(torch.nn.Linear(od, 64), torch.nn.Tanh(), torch.nn.Linear(64, 1)).cuda()
    obs_norm, rew_norm = MeanStdNormalizer(shape=(od,), clip=10.0), RewardStdNormalizer(gamma=0.99, clip=10.0)
    buf = PPOBuffer(env.observation_space, env.action_space, T, n)
    zeros = torch.zeros(n, 1, device='cuda')

    def rollout(obs):
        with torch.no_grad():
            for _ in range(T):
                act = actor(obs) + 0.1 * torch.randn(n, nu, device='cuda')          # [N, nu] rows: read in place by the kernel
                v = critic(obs)
                next_obs, rew, done, info = env.step(act)
                next_obs = obs_norm(next_obs)               # a fresh tensor per call: `obs` of this step stays intact
                rew = rew_norm(rew, done)
                terminal_v = torch.where(info['TimeLimit.truncated'][:, None], critic(info['terminal_observation']), zeros)
                buf.push({'obs': obs, 'act': act, 'rew': rew, 'mask': 1.0 - done.float(), 'v': v, 'terminal_v': terminal_v})
                obs = next_obs
            ret, adv = compute_returns_and_advantages(buf.rew, buf.v, buf.mask, buf.terminal_v, critic(obs), gamma=0.99,
                                                      use_gae=True, gae_lambda=0.95)
        return obs, ret, adv

    obs = obs_norm(env.reset())
    obs, _, _ = rollout(obs)                                                        # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 5
    for _ in range(iters):
        obs, ret, adv = rollout(obs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%d envs x %d steps x %d rollouts in %.3f s: %.3e env-steps/s through policy + env + normalisers + buffer + GAE'
          % (n, T, iters, dt, n * T * iters / dt))
    print('advantage mean %.3e std %.3e; return mean %.3e' % (float(adv.mean()), float(adv.std()), float(ret.mean())))
    env.close()


if __name__ == '__main__':
    main()
