#!/usr/bin/env python
"""The rollout half of PPO (reference: controllers/ppo/ppo.py:247-293) with everything on the device and nothing copied
between the stages: policy -> env.step -> observation / reward normalisers -> time-limit bootstrap -> PPOBuffer -> returns + GAE.

``PPOBuffer(..., env=env)`` lays the buffer out as the engine's own planes: the step kernel writes ``done`` into row t, the
normalisers' kernels write the normalised observation / reward into the buffer's planes (``out=``), ``push`` carries only what
the policy produces, and GAE reads the done bytes and the planes in place.  The policy is a small random MLP (no learning):
the point is the data path.

    python examples/ppo_rollout.py [num_envs] [rollout_steps] [pushed|planes|graph]

``graph``: the plane-layout rollout with its T steps recorded once in a HIP graph (``env.capture_loop``: MLP kernels, step kernel,
normalisers; the step serial through a device word, so every replay draws fresh resets) and replayed per rollout -- no host in the
loop, which is what bounds the small batches.
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
    mode = sys.argv[3] if len(sys.argv) > 3 else 'planes'
    planes, graph = mode in ('planes', 'graph'), mode == 'graph'
    env = scg.make('quadrotor', quad_type=2, num_envs=n, device='cuda', seed=0, task='traj_tracking', cost='rl_reward',
                   normalized_rl_action_space=True, obs_goal_horizon=1, randomized_init=True, copy_outputs=not planes)
    # (planes: the normaliser consumes the raw observation at once and writes into the buffer, so the env may step in place)
    od, nu = env.obs_dim, env.action_dim
    torch.manual_seed(0)
    actor = torch.nn.Sequential(torch.nn.Linear(od, 64), torch.nn.Tanh(), torch.nn.Linear(64, nu), torch.nn.Tanh()).cuda()
    critic = torch.nn.Sequential(torch.nn.Linear(od, 64), torch.nn.Tanh(), torch.nn.Linear(64, 1)).cuda()
    obs_norm, rew_norm = MeanStdNormalizer(shape=(od,), clip=10.0), RewardStdNormalizer(gamma=0.99, clip=10.0)
    buf = PPOBuffer(env.observation_space, env.action_space, T, n, env=env if planes else None)
    zeros = torch.zeros(n, 1, device='cuda')

    def step_body(t):                                  # one step of the plane-layout rollout, capture-safe (graph mode records T of them)
        obs = buf.obs_at(t)
        act = actor(obs) + 0.1 * torch.randn(n, nu, device='cuda')
        v = critic(obs)
        raw, rew, done, info = env.step(act, out=buf.step_out(t, obs=False, reward=False))
        obs_norm(raw, out=buf.obs_row(t + 1))
        rew_norm(rew, done, out=buf.rew_row(t))
        terminal_v = torch.where(info['TimeLimit.truncated'][:, None], critic(info['terminal_observation']), zeros)
        buf.act[t].copy_(act); buf.v[t].copy_(v); buf.terminal_v[t].copy_(terminal_v)      # (push() without its host-side ring index)

    loop = None

    def rollout(obs):
        with torch.no_grad():
            if graph:
                buf.start(obs)
                loop.replay()
                obs = buf.obs_at(T)
                ret, adv = compute_returns_and_advantages(buf.rew, buf.v, None, buf.terminal_v, critic(obs), gamma=0.99, use_gae=True,
                                                          gae_lambda=0.95, dones=buf.done)
                return obs, ret, adv
            if planes:
                obs = buf.start(obs) if obs.data_ptr() != buf.obs_at(0).data_ptr() else obs
            for t in range(T):
                act = actor(obs) + 0.1 * torch.randn(n, nu, device='cuda')          # [N, nu] rows: read in place by the kernel
                v = critic(obs)
                if planes:
                    raw, rew, done, info = env.step(act, out=buf.step_out(t, obs=False, reward=False))   # done -> row t of the buffer
                    next_obs = obs_norm(raw, out=buf.obs_row(t + 1))                                  # normalised obs -> slab t + 1
                    rew_norm(rew, done, out=buf.rew_row(t))                                           # normalised reward -> row t
                else:
                    raw, rew, done, info = env.step(act)
                    next_obs = obs_norm(raw)            # a fresh tensor per call: `obs` of this step stays intact
                    rew = rew_norm(rew, done)
                terminal_v = torch.where(info['TimeLimit.truncated'][:, None], critic(info['terminal_observation']), zeros)
                if planes:
                    buf.push({'act': act, 'v': v, 'terminal_v': terminal_v})
                else:
                    buf.push({'obs': obs, 'act': act, 'rew': rew, 'mask': 1.0 - done.float(), 'v': v, 'terminal_v': terminal_v})
                obs = next_obs
            if planes:
                ret, adv = compute_returns_and_advantages(buf.rew, buf.v, None, buf.terminal_v, critic(obs), gamma=0.99, use_gae=True,
                                                          gae_lambda=0.95, dones=buf.done)
            else:
                ret, adv = compute_returns_and_advantages(buf.rew, buf.v, buf.mask, buf.terminal_v, critic(obs), gamma=0.99,
                                                          use_gae=True, gae_lambda=0.95)
        return obs, ret, adv

    obs = obs_norm(env.reset())
    if graph:
        assert T % 2 == 0, 'graph mode: an even number of steps (the normalisers double-buffer their statistics)'
        buf.start(obs)
        with torch.no_grad():
            loop = env.capture_loop(step_body, T, warmup=2)
    obs, _, _ = rollout(obs)                                                        # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 5
    for _ in range(iters):
        obs, ret, adv = rollout(obs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%s buffer: %d envs x %d steps x %d rollouts in %.3f s: %.3e env-steps/s through policy + env + normalisers + buffer + GAE'
          % ('plane-layout buffer, rollout replayed from a HIP graph' if graph else 'plane-layout (zero-copy)' if planes else 'pushed',
             n, T, iters, dt, n * T * iters / dt))
    print('advantage mean %.3e std %.3e; return mean %.3e' % (float(adv.mean()), float(adv.std()), float(ret.mean())))
    env.close()


if __name__ == '__main__':
    main()
