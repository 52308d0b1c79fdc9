"""hipGraph replay of a rollout's launches vs plain launches, at the BASELINE sizes (C4 2^20, C3 2^18, C2 2^16) and a tiny batch.
torch.cuda.graph captures the HIP stream the C-ABI launches on (scg_rollout with ONE queue: a capture has one origin stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import safe_control_gym_amd as scg

C3_CONSTRAINTS = [{'constraint_form': 'bounded_constraint', 'constrained_variable': 'state', 'active_dims': [0, 2, 4],
                   'lower_bounds': [-2.0, 0.0, -0.5], 'upper_bounds': [2.0, 2.0, 0.5]}]
CASES = [('C4 quad3d 2^20', 'quadrotor', dict(quad_type=3, task='traj_tracking', cost='quadratic'), 1 << 20),
         ('C3 quad2d 2^18', 'quadrotor', dict(quad_type=2, task='stabilization', cost='rl_reward', constraints=C3_CONSTRAINTS), 1 << 18),
         ('C2 cartpole 2^16', 'cartpole', dict(task='stabilization', cost='quadratic', ctrl_freq=50, pyb_freq=50), 1 << 16),
         ('quad3d 2^12', 'quadrotor', dict(quad_type=3, task='traj_tracking', cost='quadratic'), 1 << 12)]
for name, kind, kw, n in CASES:
    for queues in (1, 2):
        env = scg.make(kind, num_envs=n, device='cuda', seed=1, randomized_init=True, info_level='minimal', track_episode_stats=True,
                       rollout_queues=queues, **kw)
        env.reset()
        ld, nu = env.plane_stride, env.action_dim
        lo, hi = float(env.action_space.low[0]), float(env.action_space.high[0])
        acts = torch.rand((300, nu, ld), device='cuda') * (hi - lo) + lo
        env.rollout(acts, steps=300); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout(acts, steps=1500); e1.record(); torch.cuda.synchronize()
        plain = e0.elapsed_time(e1) / 1500 * 1e3
        graph = float('nan')
        if queues == 1:
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
        print('%-18s queues %d: plain launches %.2f us/step   graph replay %.2f us/step' % (name, queues, plain, graph), flush=True)
        env.close()
