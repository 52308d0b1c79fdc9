"""Probe behind the two launch queues of scg_rollout: one env of 2^20 vs the same batch cut in 2 / 3 envs whose rollouts are issued
on separate HIP streams (the envs are made with rollout_queues=1 so that scg_rollout itself does not split).  Measured: 29.1 / 25.7 / 31.8 us
per step of the whole batch for 1 / 2 / 3 streams (and 29.9 for 4)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_control_gym_amd as scg

def mk(n, base):
    e = scg.make('quadrotor', quad_type=3, num_envs=n, device='cuda', seed=1337, env_index_base=base, randomized_init=True,
                 task='traj_tracking', cost='quadratic', info_level='minimal', track_episode_stats=True, rollout_queues=1)
    e.reset()
    lo, hi = float(e.action_space.low[0]), float(e.action_space.high[0])
    a = torch.rand((300, 4, e.plane_stride), device='cuda') * (hi - lo) + lo
    return e, a

N = 3 * (1 << 18) * 1 + (1 << 18)
full, af = mk(N, 0)
full.rollout(af, steps=300); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); full.rollout(af, steps=1500); e1.record(); torch.cuda.synchronize()
print('one kernel per step        : %.2f us/step' % (e0.elapsed_time(e1) / 1500 * 1e3))
for parts in (2, 3):
    envs = [mk(N // parts, k * (N // parts)) for k in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for (e, a), s in zip(envs, streams):
        with torch.cuda.stream(s):
            e.rollout(a, steps=100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # interleave the issue order step by step, as one C loop over `parts` streams would
    for k in range(15):
        for (e, a), s in zip(envs, streams):
            with torch.cuda.stream(s):
                e.rollout(a, steps=100)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%d streams x %d envs, 100-step chunks: %.2f us per step of the whole batch' % (parts, N // parts, dt / 1500 * 1e6))
    for e, _ in envs: e.close()
