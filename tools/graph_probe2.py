"""Round 3: scg_rollout's own hipGraph replay (scg_config.rollout_graph) against plain launches and against a caller-side
torch.cuda.graph capture, at small batches.  us per step; events around `reps` back-to-back rollouts of `steps` steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_control_gym_amd as scg

CASES = [('C2 cartpole 2^16', 'cartpole', dict(task='stabilization', cost='quadratic', ctrl_freq=50, pyb_freq=50), 1 << 16),
         ('quad3d 2^12', 'quadrotor', dict(quad_type=3, task='traj_tracking', cost='quadratic'), 1 << 12),
         ('quad3d 2^16', 'quadrotor', dict(quad_type=3, task='traj_tracking', cost='quadratic'), 1 << 16)]


def timed(fn, reps, steps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(400000)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * steps) * 1e3


for name, kind, kw, n in CASES:
    for steps in (100, 300, 1000):
        reps = max(1, 1500 // steps)
        row = []
        for mode in ('plain', 'internal', 'torch'):
            env = scg.make(kind, num_envs=n, device='cuda', seed=1, randomized_init=True, info_level='minimal', track_episode_stats=True,
                           rollout_queues=1, rollout_graph=(0 if mode == 'internal' else -1), **kw)
            env.reset()
            ld, nu = env.plane_stride, env.action_dim
            lo, hi = float(env.action_space.low[0]), float(env.action_space.high[0])
            acts = torch.rand((300, nu, ld), device='cuda') * (hi - lo) + lo
            for _ in range(3):
                env.rollout(acts, steps=steps)
            if mode == 'torch':
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    env.rollout(acts, steps=10)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=s):
                        env.rollout(acts, steps=steps)
                torch.cuda.synchronize()
                g.replay()
                us = min(timed(g.replay, reps, steps) for _ in range(3))
            else:
                us = min(timed(lambda: env.rollout(acts, steps=steps), reps, steps) for _ in range(3))
            if mode == 'internal':
                assert env._engine.rollout_graph_info()[0] == 1
            row.append('%s %.2f' % (mode, us))
            env.close()
        print('%-18s steps %4d x %2d: %s   us/step' % (name, steps, reps, '   '.join(row)), flush=True)
