import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_control_gym_amd as scg
for n in (1 << 20, 1 << 16, 1 << 12):
    env = scg.make('quadrotor', quad_type=3, num_envs=n, device='cuda', seed=1, randomized_init=True, task='traj_tracking',
                   cost='quadratic', info_level='minimal', track_episode_stats=True)
    env.reset()
    ld = env.plane_stride
    lo, hi = float(env.action_space.low[0]), float(env.action_space.high[0])
    acts = torch.rand((300, 4, ld), device='cuda') * (hi - lo) + lo
    env.rollout(acts, steps=300); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env.rollout(acts, steps=1500); e1.record(); torch.cuda.synchronize()
    plain = e0.elapsed_time(e1) / 1500 * 1e3
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        env.rollout(acts, steps=10)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            env.rollout(acts, steps=300)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    graph = e0.elapsed_time(e1) / 1500 * 1e3
    print('n=%d  plain launches %.2f us/step   graph replay %.2f us/step' % (n, plain, graph), flush=True)
    env.close()
