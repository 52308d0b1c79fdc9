"""Probe (round 3): Q launch queues for the rollout, emulated with Q envs of 2^20 / Q envs each, every one rolling out on a stream of its own
(one queue per env: scg_config.rollout_queues = 1).  Q = 2 is what scg_rollout does inside one handle."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import safe_control_gym_amd as scg
N, steps = 1 << 20, 300
def run(Q):
    n = N // Q // 256 * 256
    envs, seqs, streams = [], [], []
    for q in range(Q):
        e = scg.make('quadrotor', quad_type=3, num_envs=n, device='cuda', seed=1 + q, randomized_init=True, task='traj_tracking', cost='quadratic',
                     info_level='minimal', rollout_queues=1)
        e.reset()
        lo, hi = float(e.action_space.low[0]), float(e.action_space.high[0])
        envs.append(e); seqs.append(torch.rand((64, 4, e._ld), device='cuda') * (hi - lo) + lo); streams.append(torch.cuda.Stream())
    def go():
        for e, s, st in zip(envs, seqs, streams):
            with torch.cuda.stream(st):
                e.rollout(s, steps=steps)
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); go(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e6)
    print('Q = %d queues x %7d envs: %.2f us per step of the whole batch (best of 5 x %d steps)' % (Q, n, best, steps))
    for e in envs: e.close()
for Q in (1, 2, 3, 4, 2, 3):
    run(Q)
