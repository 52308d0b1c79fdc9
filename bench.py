#!/usr/bin/env python
"""Headline benchmark: env-steps/sec of the fused HIP ``env.step()`` at N parallel envs.

Contract (see the build prompt): ``python bench.py --gpus N --steps K --warmup W`` -> rank 0 prints
ONE JSON line.  A "step" is one pass of the hot path over one batch of synthetic input: every env of the
batch advances by one control step (``scg_rollout`` issues the fused ``scg_step`` kernel once per step, or --
from 2^18 envs on -- once per half batch and step on two HIP streams; full HBM traffic every step).

Workload (BASELINE.json ``configs``) -- the SAME per-GPU work at every N, so that the
per-N values measure scaling and nothing else:
  * C4 = Quadrotor-3D trajectory tracking, 2^20 envs per GPU, quadratic cost, RK4, auto-reset, with the per-env
    episode-return tracking every vectorised caller of the reference has (VecRecordEpisodeStatistics):
    algorithmic bytes B = 8 nx + 4 nu + 13 + 16 = 141 per env-step.
  * N > 1 -> C5: that shard on every GPU (weak scaling, envs sharded by contiguous global index) plus one RCCL
    all-gather of the episode returns every 300 steps (= one episode) inside the timed region.
  * At N = 1 the line also carries ``c4_without_episode_stats``: the bare 125-byte step (``--no-episode-stats``
    makes it the primary measurement).
Actions are i.i.d. U(a_low, a_high) float32, pre-generated on the device ([300, nu, N]) and reused
cyclically; initial / reset states are nominal + the reference's uniform offsets (Philox on device).
Inputs are resident in HBM when the timed region starts.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (6.3 TB/s achievable)


def build_env(args, rank, world):
    import safe_control_gym_amd as scg
    n = args.envs_per_gpu
    track = not args.no_episode_stats
    common = dict(num_envs=n, device='cuda', seed=1337, env_index_base=rank * n, randomized_init=True,
                  info_level='minimal', track_episode_stats=track, ctrl_freq=60, pyb_freq=240, episode_len_sec=5)
    import yaml
    for kv in args.env_kw:
        k, v = kv.split('=', 1)
        common[k] = yaml.safe_load(v)
    w = args.workload
    if w in ('c4', 'c5'):
        env = scg.make('quadrotor', quad_type=3, task='traj_tracking', cost='quadratic', **common)
        name = 'Quadrotor-3D traj_tracking (circle), quadratic cost Q=R=I, RK4 dt=1/60, auto-reset'
    elif w == 'c3':
        cons = [{'constraint_form': 'bounded_constraint', 'constrained_variable': 'state', 'active_dims': [0, 2, 4],
                 'lower_bounds': [-2, 0, -0.5], 'upper_bounds': [2, 2, 0.5]}]
        env = scg.make('quadrotor', quad_type=2, task='stabilization', cost='rl_reward', constraints=cons, **common)
        name = 'Quadrotor-2D stabilization, rl_reward, BoundedConstraint(active_dims [0,2,4])'
    elif w == 'c2':
        common.update(ctrl_freq=50, pyb_freq=50)
        env = scg.make('cartpole', task='stabilization', cost='quadratic', **common)
        name = 'CartPole stabilization, quadratic cost, RK4 dt=1/50 (no reference env: parity unpinned)'
    else:
        raise SystemExit('unknown workload %r' % w)
    nx, nu = env.state_dim, env.action_dim
    bytes_per = 8 * nx + 4 * nu + 13
    if track:
        bytes_per += 16                                  # ep_return + ep_length read-modify-write (141 B for C5);
        # the engine moves 8 of them: under auto-reset the episode length IS the step counter, so no ep_length plane exists
    if env.constraints is not None:
        bytes_per += 4 * env.constraints.num_device_rows + 1
    return env, name, bytes_per


def cpu_baseline(args, env_kwargs_name):
    """The C float64 port of the same step (oracle/c), timed on this box's host cores (bounded sample)."""
    import numpy as np
    from oracle.ref_env import RefEnv
    from oracle.c_port import CPort
    w = args.workload
    if w in ('c4', 'c5'):
        ref = RefEnv(system=3, task='traj_tracking', cost='quadratic', ctrl_freq=60, pyb_freq=240)
    elif w == 'c3':
        cons = [{'constraint_form': 'bounded_constraint', 'constrained_variable': 'state', 'active_dims': [0, 2, 4],
                 'lower_bounds': [-2, 0, -0.5], 'upper_bounds': [2, 2, 0.5]}]
        ref = RefEnv(system=2, task='stabilization', cost='rl_reward', ctrl_freq=60, pyb_freq=240, constraints=cons)
    else:
        ref = RefEnv(system='cartpole', task='stabilization', cost='quadratic', ctrl_freq=50, pyb_freq=50)
    port = CPort.from_ref_env(ref)
    n = 1 << 18
    rng = np.random.default_rng(1337)
    from oracle.ref_env import INIT_STATE_LABELS
    lo = np.array([ref.INIT_STATE_RAND_INFO[k]['low'] for k in INIT_STATE_LABELS[ref.system]])
    hi = np.array([ref.INIT_STATE_RAND_INFO[k]['high'] for k in INIT_STATE_LABELS[ref.system]])
    x0 = rng.uniform(lo, hi, size=(n, ref.nx))
    state, t = x0.copy(), np.zeros(n, dtype=np.int32)
    acts = rng.uniform(ref.action_space_low, ref.action_space_high, size=(8, n, ref.nu))
    threads = port.max_threads()
    port.step(acts[0], state, t, x0, num_threads=threads)                   # warm-up (page faults, thread pool)
    steps, t0 = 0, time.perf_counter()
    while True:
        port.step(acts[steps % 8], state, t, x0, num_threads=threads)
        steps += 1
        el = time.perf_counter() - t0
        if el > args.cpu_seconds or steps >= 400:
            break
    # one-core Python oracle (the structure-equivalent stand-in for the reference's own step()): tiny sample
    from oracle.ref_env import RefEnv as _R
    ref.reset_sampler = lambda i, e: np.zeros(ref.nx)
    ref.RANDOMIZED_INIT = False
    ref.reset()
    k, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < 2.0:
        o, r, d, _ = ref.step(acts[0][k % n])
        if d:
            ref.reset()
        k += 1
    py_rate = k / (time.perf_counter() - t1)
    return {'value': steps * n / el, 'unit': 'env-steps/s', 'cores': threads, 'kind': 'port',
            'sample': 'oracle/c float64 OpenMP port, %d envs x %d steps (%.1f s) of the same workload; '
                      'python oracle 1 core: %.0f env-steps/s' % (n, steps, el, py_rate)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3000)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--envs-per-gpu', type=int, default=1 << 20)
    ap.add_argument('--workload', default=None, help='c4 (default; called c5 when sharded over > 1 GPU), c3, c2')
    ap.add_argument('--episode-stats', action='store_true', help='(default now) track episode returns (B += 16)')
    ap.add_argument('--no-episode-stats', action='store_true', help='bare step without episode-return tracking (1 GPU only)')
    ap.add_argument('--action-period', type=int, default=300)
    ap.add_argument('--python-loop', action='store_true', help='also time env.step() called from Python per step')
    ap.add_argument('--fused', action='store_true', help='also time the one-launch multi-step rollout (own byte count)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-bare', action='store_true', help='skip the secondary c4_without_episode_stats measurement')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--env-kw', action='append', default=[], help='extra make() kwargs key=value (yaml), for experiments')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from safe_control_gym_amd import parallel

    rank, world, local_rank = parallel.init_from_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`'
                             % (args.gpus, args.gpus))
    if args.workload is None:
        args.workload = 'c4' if world == 1 else 'c5'        # same env; 'c5' = sharded + all-gather
    if world > 1 and args.no_episode_stats:
        raise SystemExit('--no-episode-stats is a single-GPU option (the sharded run gathers the episode returns)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    env, wname, bytes_per = build_env(args, rank, world)
    n, nu = env.num_envs, env.action_dim
    ld = env.plane_stride
    g = torch.Generator(device=dev)
    g.manual_seed(1337 + rank)
    lo = torch.as_tensor(env.action_space.low, device=dev)[None, :, None]
    hi = torch.as_tensor(env.action_space.high, device=dev)[None, :, None]
    actions = torch.rand((args.action_period, nu, ld), generator=g, device=dev, dtype=torch.float32) * (hi - lo) + lo
    env.reset()
    gather_every = 300

    def run(steps):
        done_steps = 0
        while done_steps < steps:
            chunk = min(gather_every, steps - done_steps) if world > 1 else steps - done_steps
            env.rollout(actions, steps=chunk)
            done_steps += chunk
            if world > 1:
                parallel.all_gather_episode_returns(env.episode_return, total_envs=n * world)

    run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(args.steps)
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    dev_ms = ev0.elapsed_time(ev1)
    t = torch.tensor([wall_ms, dev_ms], dtype=torch.float64, device=dev if (world == 1 or dist.get_backend() == 'nccl') else 'cpu')
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_ms, dev_ms = float(t[0]), float(t[1])
    ms_per_step = wall_ms / args.steps
    value = n * world * args.steps / (wall_ms * 1e-3)

    py_value = None
    if args.python_loop and world == 1:
        for k in range(20):
            env.step(actions[k % args.action_period])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(args.steps):
            env.step(actions[k % args.action_period])
        torch.cuda.synchronize()
        py_value = n * args.steps / (time.perf_counter() - t1)

    fused = None
    if args.fused and world == 1:
        env.rollout(actions, steps=50, fused=True)
        torch.cuda.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record()
        env.rollout(actions, steps=args.steps, fused=True)
        f1.record()
        torch.cuda.synchronize()
        fms = f0.elapsed_time(f1)
        fbytes = 4 * nu                                  # per env-step: the action plane only (state stays in registers)
        fused = {'value': n * args.steps / (fms * 1e-3), 'unit': 'env-steps/s', 'us_per_step': fms * 1e3 / args.steps,
                 'bytes_per_env_step': fbytes, 'achieved_GBs': fbytes * n * args.steps / (fms * 1e-3) / 1e9,
                 'note': 'scg_rollout_fused: ONE launch for all steps, envs stay in registers; not comparable with the '
                         'per-step roofline above (its algorithmic bytes are 4*nu per env-step); VALU-bound'}

    bare = None
    if world == 1 and not args.no_episode_stats and not args.no_bare and args.workload == 'c4' and not args.env_kw:
        # the same step without episode-return tracking (B = 125): the bare C4 roofline number
        args.no_episode_stats = True
        env2, _, bytes2 = build_env(args, rank, world)
        args.no_episode_stats = False
        env2.reset()
        env2.rollout(actions, steps=args.warmup)
        torch.cuda.synchronize()
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0.record()
        env2.rollout(actions, steps=args.steps)
        b1.record()
        torch.cuda.synchronize()
        bus = b0.elapsed_time(b1) * 1e3 / args.steps
        bare = {'value': n / (bus * 1e-6), 'unit': 'env-steps/s', 'bytes_per_env_step': bytes2, 'kernel_us_per_launch': bus,
                'achieved_GBs': bytes2 * n / (bus * 1e-6) / 1e9, 'frac': bytes2 * n / (bus * 1e-6) / 1e9 / HBM_PEAK_GBS}
        env2.close()

    two_queues = (n // 2) // 256 * 256 >= 131072 and os.environ.get('SCG_ROLLOUT_QUEUES', '2') != '1'
    if rank == 0:
        kernel_us = dev_ms * 1e3 / args.steps            # HIP-event time over the timed region / steps (passes over the batch)
        achieved = bytes_per * n / (kernel_us * 1e-6) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
        if os.path.exists(tpath):
            with open(tpath) as f:
                rec = json.load(f).get(args.workload if args.no_episode_stats else 'c5', {}) if args.workload in ('c4', 'c5') else {}
            if rec.get('envs', 1 << 20) == n and not args.env_kw:       # counters were collected at this size only
                traffic = rec.get('bytes_per_launch')
        out = {
            'metric': 'env-steps/sec at N parallel envs, 1/2/4/8 GPUs; % HBM-BW roofline',
            'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: %s%s' % (args.workload.upper(), wname,
                                                  '' if args.no_episode_stats else ' + per-env episode-return tracking'),
                       'envs_per_gpu': n,
                       'total_envs': n * world, 'bytes_per_env_step': bytes_per,
                       'moved_bytes_per_env_step': bytes_per - (8 if not args.no_episode_stats else 0),
                       'launch': ('scg_rollout: per step one fused scg_step kernel per HALF batch, the two halves advancing on two HIP '
                                  'streams so that one kernel\'s ramp-up / drain overlaps the other\'s body; time per step = time per '
                                  'pass over the whole batch') if two_queues else
                                 'one fused scg_step kernel per step (scg_rollout issues the launches)',
                       'launches_per_step': 2 if two_queues else 1,
                       'sharding': 'envs block-sharded by global index; all-gather of episode returns every '
                                   '%d steps' % gather_every if world > 1 else 'single GPU'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel_us_per_launch': kernel_us, 'algorithmic_bytes_per_launch': bytes_per * n,
                         'note': 'per step of the whole batch; with launches_per_step = 2 the two half-batch kernels of a step run '
                                 'concurrently, each for about this long, and bytes are those of both'},
        }
        if bare is not None:
            out['c4_without_episode_stats'] = bare
        if py_value is not None:
            out['python_step_api_value'] = py_value
        if fused is not None:
            out['fused_rollout'] = fused
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, wname)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
