#!/usr/bin/env python
"""Model queries for a model-based controller, batched: linearise the prior model at every point of the reference trajectory,
discretise each (A, B) pair exactly (matrix exponential on the matrix cores), and predict one step with the linear models.

The reference does this one state at a time on the host (controllers/mpc/mpc.py:170-175, lqr_utils.py:19-37 with
symbolic_systems.py:77-112 and mpc_utils.py:24-54); here the whole trajectory is one launch per query.

    python examples/linearize_along_trajectory.py [copies]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import safe_control_gym_amd as scg


def main():
    copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1024            # the reference trajectory, `copies` times (a sweep of MPC problems)
    kw = dict(quad_type=3, device='cuda', task='traj_tracking', cost='quadratic', info_level='minimal', ctrl_freq=60, pyb_freq=240)
    rows = scg.make('quadrotor', num_envs=1, **kw).X_GOAL.shape[0]     # 300 reference points (5 s at 60 Hz)
    env = scg.make('quadrotor', num_envs=rows * copies, **kw)
    env.reset()
    x_ref = torch.as_tensor(env.X_GOAL, dtype=torch.float32, device='cuda').repeat(copies, 1)         # [rows * copies, 12]
    u_ref = torch.as_tensor(env.U_GOAL, dtype=torch.float32, device='cuda').expand(x_ref.shape[0], -1).contiguous()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    A, B = env.linearize(x_ref, u_ref)                                   # df_func: [N, 12, 12], [N, 12, 4]
    Ad, Bd = env.linearize(x_ref, u_ref, discrete='exact')               # expm([[A, B], [0, 0]] dt)
    x_next_lin = env.model_linear_step(x_ref + 0.01, u_ref * 1.02, x_ref, u_ref)      # fd_linear_func
    x_next = env.symbolic.fd_func(x_ref + 0.01, u_ref * 1.02)['xf']      # the non-linear model, RK4
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = x_ref.shape[0]
    gap = float((x_next_lin - x_next).abs().max())
    print('%d operating points: Jacobians, exact discretisation, linear and non-linear one-step predictions in %.1f ms '
          '(%.2e points/s)' % (n, dt * 1e3, n / dt))
    print('max |linear - non-linear| one-step prediction 1 cm / 2 %% thrust off the reference: %.2e' % gap)
    print('Ad[0] diagonal', [round(float(v), 4) for v in Ad[0].diagonal()[:6]], ' |Bd[0]| max %.3e' % float(Bd[0].abs().max()))
    env.close()


if __name__ == '__main__':
    main()
