"""Probe (round 3): what about a rollout slows a later Python step loop?"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import safe_control_gym_amd as scg
n, steps = 1 << 20, 600
def timed(fn, k):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6
def trial(tag, queues, rollout_steps):
    kw = {} if queues is None else {'rollout_queues': queues}
    env = scg.make('quadrotor', quad_type=3, num_envs=n, device='cuda', seed=1, randomized_init=True, task='traj_tracking', cost='quadratic', info_level='minimal', **kw)
    env.reset()
    lo, hi = float(env.action_space.low[0]), float(env.action_space.high[0])
    seq = torch.rand((64, 4, env._ld), device='cuda') * (hi - lo) + lo
    k = [0]
    def one():
        env.step(seq[k[0] % 64]); k[0] += 1
    a = timed(one, steps)
    if rollout_steps:
        env.rollout(seq, steps=rollout_steps)
    b = timed(one, steps)
    c = timed(one, steps)
    print('%-44s loop %.2f | after %.2f | again %.2f us' % (tag, a, b, c))
    env.close()
trial('no rollout in between', None, 0)
trial('one-queue rollout of 1500 steps in between', 1, 1500)
trial('two-queue rollout of 1500 steps in between', 2, 1500)
trial('two-queue rollout of 2 steps in between', 2, 2)
trial('no rollout in between', None, 0)
