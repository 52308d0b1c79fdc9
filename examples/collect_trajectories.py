#!/usr/bin/env python
"""An open-loop data-collection run with every step's outputs kept on the device: what the reference's API walk-through does per
step on the host (tests/scripts/verbose_api.py:71-108 -- step, print obs / reward / done / constraint values, reset on done) for one
env, here for a batch in ONE call.

``env.rollout(actions, out=env.trajectory_buffer(T))`` writes observation / reward / done (and the constraint values) of step k into row
k of time-major planes (``scg_rollout_trajectory``): per-step launches on two queues for large batches, ``fused=True`` -- one launch,
the envs stay in registers, only the outputs stream out -- for the smaller ones (no launch per step).  Envs that finish auto-reset
inside the kernel; the row of a finishing env holds its reset observation, as the reference's vector envs return it.

    python examples/collect_trajectories.py [num_envs] [steps] [fused: 0 | 1]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import safe_control_gym_amd as scg


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    fused = bool(int(sys.argv[3])) if len(sys.argv) > 3 else n < (1 << 19)
    constraints = [{'constraint_form': 'bounded_constraint', 'constrained_variable': 'state', 'active_dims': [0, 2, 4],
                    'lower_bounds': [-2, 0, -0.5], 'upper_bounds': [2, 2, 0.5]}]          # the schema of competition/getting_started.yaml:128-156
    env = scg.make('quadrotor', quad_type=2, num_envs=n, device='cuda', seed=1337, task='stabilization', cost='rl_reward',
                   constraints=constraints, randomized_init=True, info_level='minimal', ctrl_freq=60, pyb_freq=240, episode_len_sec=2)
    env.reset()
    lo, hi = float(env.action_space.low[0]), float(env.action_space.high[0])
    g = torch.Generator(device='cuda').manual_seed(0)
    actions = torch.rand((T, env.action_dim, env.plane_stride), device='cuda', generator=g) * (hi - lo) + lo       # action_space.sample()
    traj = env.trajectory_buffer(T)
    env.rollout(actions, out=traj, fused=fused)                        # warm-up (and the first T steps of the data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        env.rollout(actions, out=traj, fused=fused)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('%d envs x %d steps with every step kept (%s): %.2f ms per rollout, %.3e env-steps/s'
          % (n, T, 'one fused launch' if fused else 'one launch per step', dt * 1e3, n * T / dt))
    obs, rew, done, cval = traj.obs, traj.reward, traj.done, traj.constraint_values        # [T, N, 6], [T, N], [T, N], [T, N, 6] views
    print('dataset: obs %s reward %s done %s constraint_values %s' % tuple(tuple(t.shape) for t in (obs, rew, done, cval)))
    k = int(done[:, 0].float().argmax()) if bool(done[:, 0].any()) else T - 1
    print('env 0, step %d: obs %s reward %.4f done %s worst constraint value %.3f'
          % (k, [round(v, 3) for v in obs[k, 0].tolist()], float(rew[k, 0]), bool(done[k, 0]), float(cval[k, 0].max())))
    print('episodes finished: %d; mean reward %.4f; steps in violation %.2f %%'
          % (int(done.sum()), float(rew.mean()), 100.0 * float(traj.constraint_violation.float().mean())))
    env.close()


if __name__ == '__main__':
    main()
