#!/usr/bin/env python
"""The step API with a policy in the loop (round 4): Python ``env.step()`` loop against the pipelined per-shard
``step_async`` / ``step_wait`` loop of ``make_vec_envs(num_pipelines=P)``, lean (views) and default ``make()`` tiers, 2^20 envs.
The policy is a device-side affine map of the observation (one elementwise kernel writing the action planes the step kernel
reads): a real data dependency between a step and the next.  Prints us per step of the WHOLE batch, host clock (sync -> K
steps -> sync) and HIP-event clock.  usage: step_api_probe.py [--envs N] [--steps K] [--tiers lean,default] [--modes ...]"""
import argparse
import json
import os
import sys
import time
from functools import partial

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import safe_control_gym_amd as scg  # noqa: E402
from safe_control_gym_amd.envs.env_wrappers.vectorized_env import make_vec_envs  # noqa: E402

TIERS = {'lean': dict(info_level='minimal'), 'default': dict(), 'ring4': dict(output_ring=4)}


class AffinePolicy:
    """act = mid + gain * obs[:nu] (clipped by the env itself), written as SoA action planes [nu, ld] in ONE kernel."""

    def __init__(self, env):
        self.nu, self.n = env.action_dim, env.num_envs
        dev = env.device
        lo = torch.as_tensor(env.action_space.low, device=dev, dtype=torch.float32)
        hi = torch.as_tensor(env.action_space.high, device=dev, dtype=torch.float32)
        self.bias = (0.5 * (lo + hi)).reshape(-1, 1)
        self.gain = (0.02 * (hi - lo)).reshape(-1, 1)
        self.act = torch.zeros((self.nu, env.plane_stride), dtype=torch.float32, device=dev)
        self.out = self.act[:, :self.n]

    def __call__(self, obs):
        # obs is [n, obs_dim], a transposed view of SoA planes: obs.T[:nu] are nu planes
        torch.addcmul(self.bias, obs.T[:self.nu], self.gain, out=self.out)
        return self.act


def timed(loop, steps, warmup, graph=0):
    loop(warmup)
    if graph:
        # the host out of the loop: `graph` steps of the loop captured once (policy kernels, events, step kernels on their queues)
        # and replayed; the Philox serials repeat per replay (timing only)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            loop(2)
            with torch.cuda.graph(g, stream=side):
                loop(graph)
        torch.cuda.synchronize()
        body, reps = loop, max(1, steps // graph)
        loop = lambda k: [g.replay() for _ in range(max(1, k // graph))]
        steps = reps * graph
        loop(graph)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    loop(steps)
    ev1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e6
    return wall, ev0.elapsed_time(ev1) / steps * 1e3


def plain_loop(n, kw, policy_on=True):
    env = scg.make('quadrotor', num_envs=n, **kw)
    pol = AffinePolicy(env)
    state = {'obs': env.reset()}
    fixed = pol(state['obs']).clone()

    def loop(k):
        obs = state['obs']
        for _ in range(k):
            obs = env.step(pol(obs) if policy_on else fixed)[0]
        state['obs'] = obs
    return loop, env.close


def pipelined_loop(n, kw, pipelines, on_queue):
    venv = make_vec_envs(partial(scg.make, 'quadrotor', **kw), batch_size=n, num_pipelines=pipelines)
    P = venv.num_pipelines
    pols = [AffinePolicy(e) for e in venv.envs]
    obs = [venv.reset(pipeline=p)[0] for p in range(P)]
    streams = [venv.pipeline_stream(p) for p in range(P)] if on_queue else None
    torch.cuda.synchronize()
    pending = [False] * P

    def loop(k):
        if on_queue:
            cur = torch.cuda.current_stream()
            for s in streams:
                s.wait_stream(cur)
        for _ in range(k):
            for p in range(P):
                if on_queue:
                    torch.cuda.set_stream(streams[p])
                if pending[p]:
                    obs[p] = venv.envs[p].step_wait()[0]
                venv.envs[p].step_async(pols[p](obs[p]))
                pending[p] = True
        for p in range(P):
            if on_queue:
                torch.cuda.set_stream(streams[p])
            obs[p] = venv.envs[p].step_wait()[0]
            pending[p] = False
        if on_queue:
            torch.cuda.set_stream(cur)
            for s in streams:
                cur.wait_stream(s)
    return loop, venv.close


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=1 << 20)
    ap.add_argument('--steps', type=int, default=600)
    ap.add_argument('--warmup', type=int, default=60)
    ap.add_argument('--tiers', default='lean,default')
    ap.add_argument('--modes', default='plain_nopolicy,plain,pipe2,pipe2_queue,pipe3,pipe4')
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--graph', type=int, default=0, help='capture this many steps of each loop in a HIP graph and time replays')
    ap.add_argument('--env-kw', action='append', default=[])
    a = ap.parse_args()
    import yaml
    base = dict(quad_type=3, device='cuda', seed=1337, randomized_init=True, task='traj_tracking', cost='quadratic', ctrl_freq=60,
                pyb_freq=240, episode_len_sec=5)
    for kv in a.env_kw:
        k, v = kv.split('=', 1)
        base[k] = yaml.safe_load(v)
    res = {}
    for r in range(a.rounds):
        for tier in a.tiers.split(','):
            kw = dict(base, **TIERS[tier])
            for mode in a.modes.split(','):
                if mode == 'plain':
                    loop, close = plain_loop(a.envs, kw)
                elif mode == 'plain_nopolicy':
                    loop, close = plain_loop(a.envs, kw, policy_on=False)
                else:
                    p = int(mode[4])
                    loop, close = pipelined_loop(a.envs, kw, p, mode.endswith('_queue'))
                wall, dev = timed(loop, a.steps, a.warmup, a.graph)
                close()
                torch.cuda.empty_cache()
                res.setdefault('%s/%s' % (tier, mode), []).append((wall, dev))
                print('round %d %-8s %-16s wall %.2f us/step  events %.2f us/step' % (r, tier, mode, wall, dev), flush=True)
    print(json.dumps({k: {'wall_us': [round(x[0], 2) for x in v], 'event_us': [round(x[1], 2) for x in v]} for k, v in res.items()}))


if __name__ == '__main__':
    main()
