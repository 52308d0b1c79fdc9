"""Size-independent properties checked at BASELINE.json's full sizes (2^20 envs) on the GPU, where the
float64 oracle is too slow to follow: fixed points, time limits, determinism, shard invariance, batch sums."""
import numpy as np
import pytest
import torch

import safe_control_gym_amd as scg

pytestmark = pytest.mark.gpu
N = 1 << 20


def _hover_env(**kw):
    x0 = np.zeros(12)
    x0[4] = 1.0
    return scg.make('quadrotor', quad_type=3, num_envs=N, device='cuda', seed=1337, ctrl_freq=60, pyb_freq=240,
                    randomized_init=False, init_state=x0, **kw)


def test_hover_is_a_fixed_point_and_time_limit_truncates_at_300(gpu):
    """Level attitude + u = m g / 4 is an equilibrium (quadrotor.py:262,328-329): 1M envs stay put for a whole
    episode, every one is truncated at step 300 exactly (benchmark_env.py:460-462) with episode length 300."""
    env = _hover_env(task='stabilization', cost='rl_reward',
                     task_info={'stabilization_goal': [0, 0, 1], 'stabilization_goal_tolerance': 0.05})
    env.reset()
    a = torch.tensor(env.U_GOAL, dtype=torch.float32, device='cuda').repeat(N, 1)
    for k in range(300):
        obs, rew, done, info = env.step(a)
        if k in (0, 150, 298):
            assert not bool(done.any())
            assert float((obs[:, 4] - 1.0).abs().max()) < 2e-5 and float(obs[:, [0, 2, 6, 7, 8]].abs().max()) < 1e-6
            assert float((rew - 1.0).abs().max()) < 1e-4            # exp(-dist) with dist ~ 0
    assert bool(done.all()) and bool(info['TimeLimit.truncated'].all())
    assert bool((info['episode']['l'] == 300).all())
    assert float((info['episode']['r'] - 300.0).abs().max()) < 0.05
    assert int(env.ctrl_step_counter.max()) == 0                   # auto-reset


def test_full_size_invariances_of_the_dynamics(gpu):
    """2^20 envs, size-independent properties of the model (quadrotor.py:624-674): positions never feed back into the
    dynamics, and the system is autonomous.  (i) Shifting every env's (x, y, z) leaves velocities, attitude and rates BIT-identical
    over 60 steps and the positions shifted by the same amount; (ii) starting the step counter at 100 instead of 0 leaves the
    whole state bit-identical (stabilisation: the counter only feeds the time limit)."""
    kw = dict(quad_type=3, num_envs=N, device='cuda', seed=21, randomized_init=True, task='stabilization', cost='rl_reward',
              task_info={'stabilization_goal': [0, 0, 1], 'stabilization_goal_tolerance': 0.05}, done_on_out_of_bound=False,
              episode_len_sec=10, info_level='minimal')
    a_env, b_env, c_env = scg.make('quadrotor', **kw), scg.make('quadrotor', **kw), scg.make('quadrotor', **kw)
    a_env.reset(), b_env.reset(), c_env.reset()
    shift = torch.tensor([0.5, -0.25, 0.125], device='cuda')
    b_env.state[:, [0, 2, 4]] += shift
    c_env.ctrl_step_counter.fill_(100)
    g = torch.Generator(device='cuda'); g.manual_seed(4)
    lo, hi = float(a_env.action_space.low[0]), float(a_env.action_space.high[0])
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    acts = torch.rand((6, 4, a_env.plane_stride), generator=g, device='cuda') * (0.2 * half) + (mid - 0.1 * half)   # near hover
    for env in (a_env, b_env, c_env):
        env.rollout(acts, steps=60)
    torch.cuda.synchronize()
    rest = [1, 3, 5, 6, 7, 8, 9, 10, 11]
    assert torch.isfinite(a_env.state).all()
    assert torch.equal(a_env.state[:, rest], b_env.state[:, rest])
    assert float(((b_env.state[:, [0, 2, 4]] - a_env.state[:, [0, 2, 4]]) - shift).abs().max()) < 2e-5
    assert torch.equal(a_env.state, c_env.state) and int(c_env.ctrl_step_counter.min()) == 160
    for env in (a_env, b_env, c_env):
        env.close()


def test_c5_total_size_eight_shards_reproduce_the_8m_batch(gpu):
    """BASELINE config C5 at its full TOTAL size on one GPU: the 2^23-env batch (one engine) against its eight 2^20-env shards
    (parallel.make_sharded: env_index_base = rank * 2^20), each run in turn on this GPU -- what the eight ranks of the 8-GPU run
    compute.  Same resets (Philox keyed by the global index), same rollout with auto-resets: states, counters and episode returns
    of every shard equal its slice of the big batch bit for bit; the gathered returns equal the big batch's plane."""
    from safe_control_gym_amd import parallel
    total, world, K = 1 << 23, 8, 40
    kw = dict(id='quadrotor', quad_type=3, device='cuda', seed=1337, ctrl_freq=60, pyb_freq=240, randomized_init=True, task='traj_tracking',
              cost='quadratic', info_level='minimal', track_episode_stats=True, episode_len_sec=0.5)
    full = parallel.make_sharded(scg.make, total, rank=0, world_size=1, **kw)
    full.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(11)
    lo, hi = float(full.action_space.low[0]), float(full.action_space.high[0])
    acts = torch.rand((3, 4, full.plane_stride), generator=g, device='cuda') * (hi - lo) + lo
    full.rollout(acts, steps=K)
    state_full, t_full, ret_full = full.state_soa.clone(), full.ctrl_step_counter.clone(), full.episode_return.clone()
    assert int((t_full < K).sum()) > total // 2                     # 30-step episodes: every env auto-reset at least once
    full.close()
    del full
    gathered = []
    for rank in range(world):
        shard = parallel.make_sharded(scg.make, total, rank=rank, world_size=world, **kw)
        lo_i, hi_i = parallel.shard_bounds(total, world, rank)
        assert shard.num_envs == hi_i - lo_i == 1 << 20
        shard.reset()
        a = torch.zeros((3, 4, shard.plane_stride), device='cuda')
        a[:, :, :shard.num_envs] = acts[:, :, lo_i:hi_i]
        shard.rollout(a, steps=K)
        assert torch.equal(shard.state_soa, state_full[:, lo_i:hi_i]), rank
        assert torch.equal(shard.ctrl_step_counter, t_full[lo_i:hi_i]), rank
        gathered.append(shard.episode_return.clone())
        shard.close()
        del shard, a
    assert torch.equal(torch.cat(gathered), ret_full)               # what the all-gather of the 8 ranks assembles


def test_lane_offsets_up_to_the_documented_limit(gpu):
    """The kernels address a plane with a wave-uniform base and ONE 32-bit byte offset per lane; the C ABI therefore admits n <= 2^30 envs
    per device (byte offsets up to 2^32 - 4).  CartPole at n = 2^30 - 192 (ragged: not a multiple of the workgroup), ~40 GB of planes:
    env.step() is ONE launch over the whole batch (offsets beyond 2^31 bytes in the upper half), rollout() the two half-batch queues.
    Small shards built with env_index_base reproduce their slices -- the start, the 2^31-byte boundary (env 2^29), the ragged end -- bit
    for bit, so every lane up there read and wrote its own env."""
    n = (1 << 30) - 192
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * (1 << 30):
        pytest.skip('needs ~60 GB of free device memory')
    kw = dict(device='cuda', seed=5, ctrl_freq=50, pyb_freq=50, randomized_init=True, task='stabilization', cost='quadratic', info_level='minimal',
              episode_len_sec=0.1)                     # 5-step episodes: auto-resets (Philox keyed by the global env index) inside the run
    big = scg.make('cartpole', num_envs=n, **kw)
    big.reset()
    ld = big.plane_stride
    acts = torch.empty((2, 1, ld), device='cuda')
    lane = torch.arange(ld, device='cuda', dtype=torch.float32)
    acts[0, 0] = torch.sin(lane * 1e-3) * 5.0
    acts[1, 0] = torch.cos(lane * 7e-4) * 5.0
    del lane
    for k in range(4):
        big.step(acts[k % 2])                          # lone launches over the whole batch
    big.rollout(acts, steps=4)                         # two queues, serials 4 .. 7
    windows = [(0, 4096), ((1 << 29) - 2048, (1 << 29) + 2048), (n - 4096, n)]
    got = [(big.state_soa[:, lo:hi].clone(), big.ctrl_step_counter[lo:hi].clone()) for lo, hi in windows]
    assert int((big.ctrl_step_counter[:1 << 20] < 8).sum()) > 0            # auto-resets happened
    big.close()
    del big
    torch.cuda.empty_cache()
    for (lo, hi), (state, t) in zip(windows, got):
        small = scg.make('cartpole', num_envs=hi - lo, env_index_base=lo, **kw)
        small.reset()
        a = torch.zeros((2, 1, small.plane_stride), device='cuda')
        a[:, :, :hi - lo] = acts[:, :, lo:hi]
        for k in range(4):
            small.step(a[k % 2])
        small.rollout(a, steps=4)
        assert torch.equal(small.state_soa, state), (lo, hi)
        assert torch.equal(small.ctrl_step_counter, t), (lo, hi)
        small.close()


def test_determinism_and_shard_invariance(gpu):
    """Same seed -> bit-identical rollouts; a shard built with env_index_base reproduces the matching slice of
    the big batch bit for bit (per-env Philox streams are keyed by the GLOBAL env index)."""
    kw = dict(quad_type=3, device='cuda', seed=7, ctrl_freq=60, pyb_freq=240, randomized_init=True,
              task='traj_tracking', cost='quadratic')
    g = torch.Generator(device='cuda')
    g.manual_seed(3)
    full = scg.make('quadrotor', num_envs=N, **kw)
    lo = torch.as_tensor(full.action_space.low, device='cuda')
    hi = torch.as_tensor(full.action_space.high, device='cuda')
    acts = [torch.rand((N, 4), generator=g, device='cuda') * (hi - lo) + lo for _ in range(12)]

    def roll(env, sl):
        env.reset()
        outs = []
        for a in acts:
            obs, rew, done, _ = env.step(a[sl])
            outs.append((obs.clone(), rew.clone(), done.clone()))
        return outs
    ref = roll(full, slice(0, N))
    again = roll(scg.make('quadrotor', num_envs=N, **kw), slice(0, N))
    for (o1, r1, d1), (o2, r2, d2) in zip(ref, again):
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    assert sum(int(d.sum()) for _, _, d in ref) > N // 8            # resets (Philox path) were exercised
    lo_i, hi_i = 3 * (N // 8), 4 * (N // 8)                         # rank 3 of 8
    shard = roll(scg.make('quadrotor', num_envs=hi_i - lo_i, env_index_base=lo_i, **kw), slice(lo_i, hi_i))
    for (o1, r1, d1), (o2, r2, d2) in zip(ref, shard):
        assert torch.equal(o1[lo_i:hi_i], o2) and torch.equal(r1[lo_i:hi_i], r2) and torch.equal(d1[lo_i:hi_i], d2)


def test_reset_randomisation_statistics(gpu):
    """Reset states = nominal + U(low, high) per component (quadrotor.py:68-134 ranges), independent streams."""
    env = scg.make('quadrotor', quad_type=3, num_envs=N, device='cuda', seed=11, randomized_init=True,
                   task='traj_tracking', cost='quadratic', ctrl_freq=60, pyb_freq=240)
    x = env.reset().double()
    lo = torch.tensor([-0.5, -0.01, -0.5, -0.01, 0.1, -0.01, -0.3, -0.3, -0.3, -0.01, -0.01, -0.01], device='cuda')
    hi = torch.tensor([0.5, 0.01, 0.5, 0.01, 1.5, 0.01, 0.3, 0.3, 0.3, 0.01, 0.01, 0.01], device='cuda')
    assert bool((x >= lo).all()) and bool((x <= hi).all())
    u = (x - lo) / (hi - lo)
    assert float((u.mean(0) - 0.5).abs().max()) < 2e-3 and float((u.std(0) - 12 ** -0.5).abs().max()) < 2e-3
    c = torch.corrcoef(u.T) - torch.eye(12, device='cuda', dtype=torch.float64)
    assert float(c.abs().max()) < 5e-3
    x2 = env.reset().double()
    assert float(((x2 - x).abs() > 0).double().mean()) > 0.99       # a new serial draws new offsets


def test_rollout_equals_step_loop_and_stats(gpu):
    """scg_rollout (K launches from one C call) == K step() calls; the kernel's wave-reduced batch sums
    agree with torch reductions."""
    kw = dict(quad_type=3, device='cuda', seed=5, ctrl_freq=60, pyb_freq=240, randomized_init=True,
              task='traj_tracking', cost='quadratic')
    n = 1 << 16
    a_env = scg.make('quadrotor', num_envs=n, **kw)
    b_env = scg.make('quadrotor', num_envs=n, **kw)
    a_env.reset(); b_env.reset()
    ld = a_env.plane_stride
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    lo = torch.as_tensor(a_env.action_space.low, device='cuda')[None, :, None]
    hi = torch.as_tensor(a_env.action_space.high, device='cuda')[None, :, None]
    acts = torch.rand((7, 4, ld), generator=g, device='cuda') * (hi - lo) + lo
    K = 20
    stripes = torch.zeros((64, 8), dtype=torch.float64, device='cuda')      # SCG_STATS_STRIPES rows, summed by the caller
    b_env._bufs['stats'] = stripes
    rew_sum, done_sum, ret_sum, len_sum = 0.0, 0, 0.0, 0
    for k in range(K):
        obs, rew, done, info = a_env.step(acts[k % 7])
        b_env.step(acts[k % 7])
        rew_sum += float(rew.double().sum()); done_sum += int(done.sum())
        ret_sum += float(info['episode']['r'][done].double().sum()); len_sum += int(info['episode']['l'][done].sum())
    assert torch.equal(a_env.state, b_env.state)
    stats = stripes.sum(0)
    assert abs(float(stats[0]) - rew_sum) <= 1e-5 * abs(rew_sum) and int(stats[1]) == done_sum
    assert abs(float(stats[3]) - ret_sum) <= 1e-5 * abs(ret_sum) and int(stats[4]) == len_sum
    c_env = scg.make('quadrotor', num_envs=n, **kw)
    c_env.reset()
    c_env.rollout(acts, steps=K)
    assert torch.equal(a_env.state, c_env.state) and torch.equal(a_env.episode_return, c_env.episode_return)
    assert torch.equal(a_env.ctrl_step_counter, c_env.ctrl_step_counter)
    # the fused multi-step launch leaves exactly the same buffers behind (lean configuration)
    kw_lean = dict(kw, info_level='minimal')
    d_env, e_env = scg.make('quadrotor', num_envs=n, **kw_lean), scg.make('quadrotor', num_envs=n, **kw_lean)
    d_env.reset(); e_env.reset()
    d_env.rollout(acts, steps=K)
    e_env.rollout(acts, steps=K, fused=True)
    torch.cuda.synchronize()
    assert torch.equal(d_env.state, e_env.state) and torch.equal(d_env.ctrl_step_counter, e_env.ctrl_step_counter)
    assert torch.equal(d_env.episode_return, e_env.episode_return) and torch.equal(d_env.episode_length, e_env.episode_length)
    assert torch.equal(d_env._bufs['reward'], e_env._bufs['reward']) and torch.equal(d_env._bufs['done'], e_env._bufs['done'])
    assert torch.equal(a_env.state, d_env.state)


def test_vec_env_protocol_and_episode_statistics(gpu):
    """make_vec_envs + VecRecordEpisodeStatistics keep the reference's call surface
    (env_wrappers/vectorized_env/__init__.py:40-65, record_episode_statistics.py:92-169)."""
    from functools import partial
    from safe_control_gym_amd.envs.env_wrappers.vectorized_env import make_vec_envs
    from safe_control_gym_amd.envs.env_wrappers.record_episode_statistics import VecRecordEpisodeStatistics
    env_func = partial(scg.make, 'quadrotor', quad_type=2, device='cuda', ctrl_freq=60, pyb_freq=240,
                       episode_len_sec=1, randomized_init=True,
                       constraints=[{'constraint_form': 'default_constraint', 'constrained_variable': 'state'}])
    venv = VecRecordEpisodeStatistics(make_vec_envs(env_func, batch_size=512, n_processes=4, seed=3), deque_size=10000)
    venv.add_tracker('constraint_violation', 0, mode='queue')
    short = VecRecordEpisodeStatistics(make_vec_envs(env_func, batch_size=512, n_processes=4, seed=3), deque_size=7)   # fewer than finish per step
    short.add_tracker('constraint_violation', 0, mode='queue')
    short.add_tracker('mse', 0, mode='accumulate')
    venv.add_tracker('mse', 0, mode='accumulate')
    short.reset()
    obs, _ = venv.reset()
    assert obs.shape == (512, 6) and venv.num_envs == 512
    a = torch.as_tensor(venv.action_space.sample(512), device='cuda')
    total_done = 0
    for k in range(70):
        obs, rew, done, info = venv.step(a)
        short.step(a)
        total_done += int(done.sum())
    assert len(venv.return_queue) == total_done == len(venv.length_queue) and total_done >= 512
    # a short queue holds what the tail of the long one holds (only its last entries cross to the host), the accumulators agree
    assert list(short.return_queue) == list(venv.return_queue)[-7:] and list(short.length_queue) == list(venv.length_queue)[-7:]
    assert list(short.queued_stats['constraint_violation']) == list(venv.queued_stats['constraint_violation'])[-7:]
    assert short.accumulated_stats['mse'] == pytest.approx(venv.accumulated_stats['mse'], rel=1e-6) and venv.accumulated_stats['mse'] > 0
    short.close()
    assert max(venv.length_queue) <= 60 and len(venv.queued_stats['constraint_violation']) == total_done
    assert venv.get_attr('CTRL_FREQ')[0] == 60 and len(venv.get_attr('CTRL_FREQ')) == 512
    per_env = info['n'][0]
    assert 'constraint_values' in per_env and 'mse' in per_env
    # the reference's per-env dict of a finished env (dummy_vec_env.py:41-47; read by ppo.py:265-269)
    i_done = int(torch.nonzero(done)[0]) if bool(done.any()) else None
    if i_done is not None:
        fin = info['n'][i_done]
        assert fin['terminal_observation'].shape == (6,) and 'mse' in fin['terminal_info'] and 'episode' in fin
        assert fin['terminal_info'].get('TimeLimit.truncated', False) == bool(info['TimeLimit.truncated'][i_done])
    venv.close()


def test_c_abi_edge_cases(gpu):
    """Empty, ragged and over-size inputs and misuse at the C boundary: status codes, never a crash."""
    import ctypes as C
    from safe_control_gym_amd import _capi
    env = scg.make('quadrotor', quad_type=3, num_envs=300, device='cuda', seed=1, task='traj_tracking', cost='quadratic',
                   ctrl_freq=60, pyb_freq=240, info_level='minimal')
    env.reset()
    lib, eng = _capi.load(), env._engine
    io = eng._bind(dict(env._bufs, reset_state=None), env._ld)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    before = env.state.clone()
    assert lib.scg_step(eng.handle, C.byref(io), 0, stream) == 0                     # empty batch: nothing happens
    torch.cuda.synchronize()
    assert torch.equal(env.state, before)
    assert lib.scg_step(eng.handle, C.byref(io), -1, stream) == -1
    assert lib.scg_step(eng.handle, C.byref(io), env._ld + 1, stream) == -1         # n > ld
    assert b'ld' in lib.scg_last_error()
    io.ld = (1 << 30) + 64
    assert lib.scg_step(eng.handle, C.byref(io), (1 << 30) + 1, stream) == -1       # beyond the 32-bit lane offsets
    io.ld = env._ld
    saved, io.reward = io.reward, None
    assert lib.scg_step(eng.handle, C.byref(io), 300, stream) == -1                 # mandatory output missing
    io.reward = saved
    io.struct_size = 8
    assert lib.scg_step(eng.handle, C.byref(io), 300, stream) == -1
    io.struct_size = C.sizeof(_capi.ScgStepIO)
    assert lib.scg_rollout(eng.handle, C.byref(io), 300, 3, 0, 0, stream) == -1     # action_period must be >= 1
    assert lib.scg_step(eng.handle, C.byref(io), 300, stream) == 0                  # and the handle still works
    torch.cuda.synchronize()
    assert not torch.equal(env.state, before)
    # a stabilisation handle refuses a reference table; a tracking one refuses too few rows for the quadratic cost
    tab = (C.c_double * (12 * 10))()
    assert lib.scg_set_reference(eng.handle, tab, 10) == -1 and b'rows' in lib.scg_last_error()
    stab = scg.make('quadrotor', quad_type=2, num_envs=4, device='cuda')
    stab.reset()
    assert lib.scg_set_reference(stab._engine.handle, tab, 10) == -1


BASELINE_CASES = [
    # BASELINE.json configs[1..3] at their full sizes (C2, C3, C4)
    ('C2_cartpole_64k', 'cartpole', 1 << 16, dict(task='stabilization', cost='quadratic', ctrl_freq=50, pyb_freq=50)),
    ('C3_quad2d_256k_bounded', 2, 1 << 18,
     dict(task='stabilization', cost='rl_reward', ctrl_freq=60, pyb_freq=240,
          constraints=[{'constraint_form': 'bounded_constraint', 'constrained_variable': 'state', 'active_dims': [0, 2, 4],
                        'lower_bounds': [-2, 0, -0.5], 'upper_bounds': [2, 2, 0.5]}])),
    ('C4_quad3d_1M_tracking', 3, 1 << 20, dict(task='traj_tracking', cost='quadratic', ctrl_freq=60, pyb_freq=240)),
]


@pytest.mark.parametrize('name,system,n,kw', BASELINE_CASES, ids=[c[0] for c in BASELINE_CASES])
def test_sampled_oracle_parity_at_baseline_sizes(gpu, name, system, n, kw):
    """The full-size batch free-runs with i.i.d. uniform actions (the benchmark's recipe); at every checked step 256
    randomly chosen envs are handed to the float64 oracle -- their state and counter BEFORE the step, the same
    action -- and next state / reward / done / constraint values / terminal observation must agree to 1e-5.
    The sample is redrawn every step, so it also lands on envs that were reset through the Philox path."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import RTOL, rel_err
    from oracle.ref_env import RefEnv
    make = (lambda **k: scg.make('cartpole', **k)) if system == 'cartpole' else (lambda **k: scg.make('quadrotor', quad_type=system, **k))
    env = make(num_envs=n, device='cuda', seed=1337, randomized_init=True, **kw)
    env.reset()
    okw = {k: v for k, v in kw.items()}
    ref = RefEnv(system=system, randomized_init=False, **okw)           # its state is overwritten before every step
    ref.reset()
    rng = np.random.default_rng(7)
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    lo = torch.as_tensor(env.action_space.low, device='cuda')
    hi = torch.as_tensor(env.action_space.high, device='cuda')
    worst, dones_seen = 0.0, 0
    for k in range(24):
        act = torch.rand((n, env.action_dim), generator=g, device='cuda') * (hi - lo) + lo
        idx = rng.choice(n, 256, replace=False)
        it = torch.as_tensor(idx, device='cuda')
        x0 = env.state[it].cpu().numpy().astype(np.float64)
        t0 = env.ctrl_step_counter[it].cpu().numpy()
        a = act[it].cpu().numpy().astype(np.float64)
        obs, rew, done, info = env.step(act)
        obs_s, rew_s, done_s = obs[it].cpu().numpy(), rew[it].cpu().numpy(), done[it].cpu().numpy()
        term = info['terminal_observation'][it].cpu().numpy()
        cv = info['constraint_values'][it].cpu().numpy() if 'constraint_values' in info else None
        for j in range(256):
            ref.state = x0[j].copy()
            ref.ctrl_step_counter = int(t0[j])
            o_r, r_r, d_r, inf_r = ref.step(a[j])
            if system == 3 and abs(ref.state[7]) >= 1.5:           # crossed the Euler-angle singularity: done only
                assert bool(done_s[j]) == bool(d_r)
                continue
            assert bool(done_s[j]) == bool(d_r), (name, k, j)
            e = abs(float(rew_s[j]) - r_r) / max(abs(r_r), 1.0)
            nxt = term[j][:len(o_r)] if d_r else obs_s[j]          # a finished env already shows its reset observation
            e = max(e, rel_err(nxt, o_r))
            if cv is not None:
                e = max(e, rel_err(cv[j], inf_r['constraint_values']))
            assert e <= RTOL, (name, k, j, e)
            worst = max(worst, e)
            dones_seen += int(d_r)
    assert dones_seen > 0
    print('%s: worst rel err %.3e over 24 x 256 sampled env-steps (%d episode ends)' % (name, worst, dones_seen))
    env.close()


def test_checkpoint_resume_is_bit_exact(gpu):
    """state_dict() / load_state_dict(): a fresh env restored from a mid-rollout snapshot continues bit for bit --
    states, rewards, dones, the Philox reset draws and the in-kernel disturbance draws (the
    reference checkpoints only the policy side and the envs' RNG states, ppo.py:115-155)."""
    import io
    n = 5000
    kw = dict(quad_type=2, num_envs=n, device='cuda', seed=9, randomized_init=True, randomized_inertial_prop=True,
              task='traj_tracking', cost='quadratic', episode_len_sec=0.5,
              disturbances={'dynamics': [{'disturbance_func': 'white_noise', 'std': 0.02}],
                            'action': [{'disturbance_func': 'impulse', 'magnitude': 0.05, 'step_offset': None, 'duration': 3}]})
    a = scg.make('quadrotor', **kw)
    a.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    lo, hi = float(a.action_space.low[0]), float(a.action_space.high[0])
    acts = [torch.rand((n, 2), generator=g, device='cuda') * (hi - lo) + lo for _ in range(50)]
    for k in range(20):
        a.step(acts[k])
    buf = io.BytesIO()
    torch.save(a.state_dict(), buf)                       # through a real serialisation round trip
    outs_a = [tuple(t.clone() for t in a.step(acts[k])[:3]) for k in range(20, 50)]
    b = scg.make('quadrotor', **dict(kw, seed=1))         # different seed: load_state_dict must bring the right one
    buf.seek(0)
    b.load_state_dict(torch.load(buf, weights_only=True))
    for k in range(20, 50):
        ob, rb, db = b.step(acts[k])[:3]
        oa, ra, da = outs_a[k - 20]
        assert torch.equal(ob, oa) and torch.equal(rb, ra) and torch.equal(db, da), k
    assert torch.equal(a.state, b.state) and torch.equal(a.episode_return, b.episode_return)
    assert int(sum(int(o[2].sum()) for o in outs_a)) > n          # episodes ended and were re-drawn after the snapshot
    a.close(), b.close()


def test_aero_terms_known_answers_on_device(gpu):
    """The ground-effect and drag terms really act in the kernel (base_aviary.py:437-500 as restated in DESIGN.md):
    level hover at 5 cm climbs with a = g * GND_EFF_COEFF (PROP_RADIUS / 4h)^2, a level vehicle drifting at v is slowed
    by k v / m with k = DRAG_COEFF * sum_i 2 pi rpm_i / 60; plain 'pyb' physics shows neither."""
    n = 256
    x0 = np.zeros(12); x0[4] = 0.05; x0[1] = 2.0                   # 5 cm above ground, drifting at 2 m/s along x
    goal = {'stabilization_goal': [0, 0, 1], 'stabilization_goal_tolerance': 0.0}
    res = {}
    for phys in ('pyb', 'pyb_gnd', 'pyb_drag', 'pyb_gnd_drag_dw'):
        env = scg.make('quadrotor', quad_type=3, num_envs=n, device='cuda', randomized_init=False, init_state=x0,
                       physics=phys, task='stabilization', cost='rl_reward', task_info=goal, ctrl_freq=60, pyb_freq=60)
        env.reset()
        a = torch.tensor(env.U_GOAL, dtype=torch.float32, device='cuda').repeat(n, 1)
        env.step(a)
        res[phys] = env.state[0].cpu().numpy().astype(np.float64)
        m, g, dt, kf = env.MASS, env.GRAVITY_ACC, env.CTRL_TIMESTEP, env.KF
        env.close()
    gain = 11.36859 * (2.31348e-2 / (4 * 0.05)) ** 2
    assert abs(res['pyb'][5]) < 1e-6 and abs(res['pyb'][1] - 2.0) < 1e-6           # equilibrium, no drag
    # one RK4 step with the thrust gain held over the step: vz = g * gain * dt
    assert res['pyb_gnd'][5] == pytest.approx(g * gain * dt, rel=2e-4) and abs(res['pyb_gnd'][1] - 2.0) < 1e-6
    srpm = 4 * np.sqrt(m * g / 4 / kf)
    k = 9.1785e-7 * 2 * np.pi / 60 * srpm
    assert res['pyb_drag'][1] - 2.0 == pytest.approx(-k * 2.0 / m * dt, rel=2e-3) and abs(res['pyb_drag'][5]) < 1e-6
    assert res['pyb_gnd_drag_dw'][5] == pytest.approx(res['pyb_gnd'][5], rel=1e-3)
    assert res['pyb_gnd_drag_dw'][1] - 2.0 == pytest.approx(res['pyb_drag'][1] - 2.0, rel=0.2)    # more thrust -> more rpm


def test_row_major_actions_are_read_in_place(gpu):
    """A policy's [N, nu] float32 output goes to the kernel as is (scg_step_io.action_rows: one vector load per lane);
    results are bit-identical to the same actions given as SoA planes or as a host array, for nu = 1, 2, 4 and for an
    env with input constraints (which judge the RAW action, constraints.py:167-168)."""
    for make_kw, nu in ((dict(quad_type=1), 1), (dict(quad_type=2), 2), (dict(quad_type=3, task='traj_tracking'), 4)):
        kw = dict(num_envs=3001, device='cuda', seed=4, randomized_init=True, cost='quadratic',
                  constraints=[{'constraint_form': 'default_constraint', 'constrained_variable': 'input'}], **make_kw)
        a, b, c = (scg.make('quadrotor', **kw) for _ in range(3))
        a.reset(), b.reset(), c.reset()
        g = torch.Generator(device='cuda'); g.manual_seed(nu)
        lo, hi = float(a.action_space.low[0]), float(a.action_space.high[0])
        for k in range(12):
            act = (torch.rand((3001, nu), generator=g, device='cuda') * 1.4 - 0.2) * (hi - lo) + lo      # some out of bounds
            planes = b.action_buffer
            planes[:, :] = act
            oa, ra, da, ia = a.step(act)                       # rows, in place
            ob, rb, db, ib = b.step(planes)                    # SoA planes
            oc, rc, dc, ic = c.step(act.cpu().numpy())         # host array (copied)
            assert a._raw_action_rows and not b._raw_action_rows and not c._raw_action_rows
            for x, y, z in ((oa, ob, oc), (ra, rb, rc), (da, db, dc), (ia['constraint_values'], ib['constraint_values'], ic['constraint_values'])):
                assert torch.equal(x, y) and torch.equal(x, z), (nu, k)
            assert torch.equal(a.current_raw_input_action, act)
        for e in (a, b, c):
            e.close()


@pytest.mark.parametrize('tier', ['lean', 'full_info', 'extra'])
def test_two_queue_rollout_is_bit_identical_to_the_step_loop(gpu, tier):
    """scg_rollout cuts batches of at least 2^19 envs (here: rollout_split_min lowered) in two halves that advance on two HIP streams (the tail of one half's
    kernel overlaps the body of the other's).  Per env nothing changes: same arithmetic, same Philox counters (global env
    index) -- so the buffers after K steps equal those of K single-stream env.step() calls bit for bit.  Odd n: the second
    half is ragged."""
    n = 300001
    kw = dict(quad_type=3, num_envs=n, device='cuda', seed=21, randomized_init=True, task='traj_tracking', cost='quadratic',
              episode_len_sec=0.25, rollout_split_min=131072)
    if tier == 'lean':
        kw.update(info_level='minimal')
    elif tier == 'extra':
        kw.update(randomized_inertial_prop=True,
                  disturbances={'dynamics': [{'disturbance_func': 'white_noise', 'std': 0.01}],
                                'observation': [{'disturbance_func': 'impulse', 'magnitude': 0.05, 'step_offset': None, 'duration': 2}]},
                  constraints=[{'constraint_form': 'default_constraint', 'constrained_variable': 'state'}])
    a, b = scg.make('quadrotor', **kw), scg.make('quadrotor', **kw)
    a.reset(), b.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(8)
    lo, hi = float(a.action_space.low[0]), float(a.action_space.high[0])
    acts = torch.rand((5, 4, a.plane_stride), generator=g, device='cuda') * (hi - lo) + lo
    K = 23
    a.rollout(acts, steps=K)
    for k in range(K):
        b.step(acts[k % 5])
    torch.cuda.synchronize()
    for key in a._bufs:
        if key == 'action':
            continue
        x, y = a._bufs[key], b._bufs[key]
        # the n live lanes / rows (the pad lanes of a freshly allocated per-step output tensor are never written)
        x, y = (x[:n], y[:n]) if (x.ndim == 1 or key == 'terminal_record') else (x[:, :n], y[:, :n])
        assert torch.equal(x, y), (tier, key)
    assert int(a.ctrl_step_counter.max()) < 15           # episodes of 15 steps: every env was reset at least once
    a.close(), b.close()


@pytest.mark.parametrize('kw', [dict(task='traj_tracking', cost='quadratic'),
                                dict(task='traj_tracking', cost='rl_reward', obs_goal_horizon=2, normalized_rl_action_space=True),
                                dict(task='stabilization', cost='rl_reward', info_level='minimal', copy_outputs=True)],
                         ids=['default', 'goal_horizon', 'minimal_fresh'])
def test_fresh_outputs_ping_pong_state(gpu, kw):
    """copy_outputs=True without observation noise: every step writes the new state into a newly allocated buffer
    (scg_step_io.state_out) that IS the observation handed out.  (i) same numbers, bit for bit, as a twin env handing out
    views; (ii) an observation, once handed out, never changes -- not by later steps, a masked reset, a full reset, a
    rollout() or a checkpoint load; (iii) env.state is the last observation's storage (nothing is moved twice)."""
    n = 3000
    base = dict(quad_type=2, num_envs=n, device='cuda', seed=11, randomized_init=True, episode_len_sec=0.4, ctrl_freq=50, pyb_freq=50)
    env = scg.make('quadrotor', **{**base, **kw})
    twin = scg.make('quadrotor', **{**base, **kw, 'copy_outputs': False})
    o0, t0 = env.reset(), twin.reset()
    assert env._pingpong and not twin._pingpong
    assert torch.equal(o0, t0)
    g = torch.Generator(device='cuda'); g.manual_seed(2)
    lo = torch.as_tensor(env.action_space.low, device='cuda'); hi = torch.as_tensor(env.action_space.high, device='cuda')
    held, dones = [(o0, o0.clone())], 0
    for k in range(60):
        a = torch.rand((n, env.action_dim), generator=g, device='cuda') * (hi - lo) + lo
        obs, rew, done, info = env.step(a)
        tobs, trew, tdone, tinfo = twin.step(a)
        assert torch.equal(obs, tobs) and torch.equal(rew, trew) and torch.equal(done, tdone)
        assert obs.data_ptr() == env.state.data_ptr()
        assert torch.equal(obs[:, :env.state_dim], env.state)
        held.append((obs, obs.clone()))
        dones += int(done.sum())
        if k == 20:
            mask = torch.zeros(n, dtype=torch.bool); mask[::3] = True
            env.reset(mask=mask); twin.reset(mask=mask)
        if k == 40:
            env.load_state_dict(env.state_dict())
    assert dones > n
    assert len({h[0].data_ptr() for h in held}) == len(held)
    ld = env.plane_stride
    acts = torch.rand((5, env.action_dim, ld), generator=g, device='cuda') * (hi - lo)[None, :, None] + lo[None, :, None]
    env.rollout(acts); twin.rollout(acts)
    assert torch.equal(env.state, twin.state)
    env.reset(); twin.reset()
    assert torch.equal(env.state, twin.state)
    for tensor, snapshot in held:
        assert torch.equal(tensor, snapshot)
    env.close(); twin.close()


def test_fresh_outputs_survive_in_place_writers_with_an_observation_disturbance(gpu):
    """copy_outputs=True WITH an observation disturbance (no ping-pong: the kernel writes a separate obs buffer): obs, rew and
    done of a step are fresh tensors, and none of them is written again by reset(mask), reset(), rollout() or
    load_state_dict() (they used to stay bound in the engine's buffer table)."""
    n = 2048
    env = scg.make('quadrotor', quad_type=2, num_envs=n, device='cuda', seed=5, randomized_init=True, episode_len_sec=0.4,
                   ctrl_freq=50, pyb_freq=50, copy_outputs=True,
                   disturbances={'observation': [{'disturbance_func': 'white_noise', 'std': 0.05}]})
    env.reset()
    assert env._fresh_outputs and not env._pingpong
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    lo = torch.as_tensor(env.action_space.low, device='cuda'); hi = torch.as_tensor(env.action_space.high, device='cuda')
    ld = env.plane_stride
    acts = torch.rand((4, env.action_dim, ld), generator=g, device='cuda') * (hi - lo)[None, :, None] + lo[None, :, None]
    sd = None

    def stepped():
        a = torch.rand((n, env.action_dim), generator=g, device='cuda') * (hi - lo) + lo
        obs, rew, done, _ = env.step(a)
        return [(t, t.clone()) for t in (obs, rew, done)]

    held = []
    for writer in ('masked reset', 'full reset', 'rollout', 'checkpoint load', 'step'):
        for _ in range(3):
            got = stepped()
        held += got
        if writer == 'masked reset':
            mask = torch.zeros(n, dtype=torch.bool); mask[::2] = True
            sd = env.state_dict()
            env.reset(mask=mask)
        elif writer == 'full reset':
            env.reset()
        elif writer == 'rollout':
            env.rollout(acts)
        elif writer == 'checkpoint load':
            env.load_state_dict(sd)
        for tensor, snapshot in held:
            assert torch.equal(tensor, snapshot), writer
    assert len({t.data_ptr() for t, _ in held}) == len(held)
    env.close()


def test_reference_attribute_surface_after_a_step(gpu):
    """Attributes callers of the reference read off the env (SURVEY 8b-5 and the assignments of benchmark_env.py /
    quadrotor.py): current_preprocessed_action (:833), cnstr_violation (:441-449), goal_reached (:965), the counters, the
    per-env OVERRIDDEN_QUAD_MASS / _INERTIA (:466-467) -- batched, against the oracle's scalars."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from helpers import make_pair, sync_reset, sync_step, force_state
    n = 40
    cons = [{'constraint_form': 'bounded_constraint', 'constrained_variable': 'state', 'active_dims': [0, 2],
             'lower_bounds': [-0.2, 0.8], 'upper_bounds': [0.2, 1.2]}]
    env, ref, sampler = make_pair(2, n, randomized_init=True, task='stabilization', cost='quadratic', ctrl_freq=60, pyb_freq=240,
                                  normalized_rl_action_space=True, constraints=cons,
                                  engine_kwargs=dict(randomized_inertial_prop=False))
    assert env.state is None and env.ctrl_step_counter == 0 and env.current_preprocessed_action is None
    sync_reset(env, ref, sampler)
    rng = np.random.default_rng(8)
    a = rng.uniform(-1.4, 1.4, size=(n, 2)).astype(np.float32)
    force_state(env, ref)
    sync_step(env, ref, sampler, a)
    want = np.stack([np.asarray(r.current_preprocessed_action) for r in ref.envs])
    assert np.abs(env.current_preprocessed_action.cpu().numpy() - want).max() <= 1e-6
    viol = np.array([bool(r.cnstr_violation) if hasattr(r, 'cnstr_violation') else False for r in ref.envs])
    got_v = env.cnstr_violation.cpu().numpy()
    assert got_v.dtype == bool and got_v.shape == (n,) and got_v.any() and not got_v.all()
    if any(hasattr(r, 'cnstr_violation') for r in ref.envs):
        assert np.array_equal(got_v, viol)
    assert env.goal_reached.shape == (n,) and int(env.ctrl_step_counter.max()) == 1
    assert env.OVERRIDDEN_QUAD_MASS == env.MASS and len(env.OVERRIDDEN_QUAD_INERTIA) == 3
    env.close()
    env = scg.make('quadrotor', quad_type=3, num_envs=n, device='cuda', seed=1, task='traj_tracking', randomized_inertial_prop=True)
    env.reset()
    assert env.OVERRIDDEN_QUAD_MASS.shape == (n,) and env.OVERRIDDEN_QUAD_INERTIA.shape == (n, 3)
    assert float(env.OVERRIDDEN_QUAD_MASS.std()) > 0
    env.close()


def test_non_finite_actions(gpu):
    """Edge case "nulls": +-inf action components clip to the thrust bounds exactly as np.clip does in the reference
    (quadrotor.py:830).  A NaN component is where the engine deviates on purpose: np.clip would hand the NaN on (state and
    reward NaN until the time limit ends the episode, since NaN compares false against every bound); the kernel's clamp is one
    v_med3_f32, which returns the LOWER bound for a NaN operand, so the env stays finite.  Pinned here: NaN == lower bound."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from helpers import make_pair, sync_reset, sync_step, force_state, rel_err
    n = 64
    env, ref, sampler = make_pair(3, n, randomized_init=True, task='traj_tracking', cost='quadratic', ctrl_freq=60, pyb_freq=240)
    sync_reset(env, ref, sampler)
    lo, hi = env.action_space.low.astype(np.float32), env.action_space.high.astype(np.float32)
    rng = np.random.default_rng(2)
    a = rng.uniform(lo, hi, size=(n, 4)).astype(np.float32)
    a[0, 1], a[1, 0], a[2, 3], a[3, :] = np.inf, -np.inf, np.nan, np.nan
    a_ref = a.copy()
    a_ref[2, 3], a_ref[3, :] = lo[3], lo                      # what the kernel reads a NaN as
    force_state(env, ref)
    sampler.serial = env._serial
    obs, rew, done, info = env.step(a)
    obs_r, rew_r, done_r, _ = ref.step(a_ref.astype(np.float64))
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert np.array_equal(done.cpu().numpy(), done_r.astype(bool))
    assert rel_err(obs.cpu().numpy(), obs_r) <= 1e-5 and rel_err(rew.cpu().numpy(), rew_r) <= 1e-5
    # a non-finite STATE (set from outside) counts as out of bounds: the env ends its episode and restarts finite, where the
    # reference would carry the NaN to the time limit ("ill-conditioned lanes")
    env.state[5, 0] = float('nan')
    env.state[6, 7] = float('inf')
    a2 = rng.uniform(lo, hi, size=(n, 4)).astype(np.float32)
    obs, rew, done, info = env.step(a2)
    assert bool(done[5]) and bool(done[6]) and torch.isfinite(obs).all()
    assert int(env.ctrl_step_counter[5]) == 0 and int(env.ctrl_step_counter[6]) == 0
    env.close()


@pytest.mark.parametrize('physics', ['pyb', 'euler'])
def test_non_finite_velocity_ends_the_episode_in_the_same_step(gpu, physics):
    """The reference bounds velocities and rates with +-float32 max (quadrotor.py:742-791): only a non-finite value violates them
    (inf > f32max).  The RK4 kernel skips those compares because a non-finite velocity reaches the bounded position within the
    same step; explicit Euler advances the position with the OLD velocity, so there every dim keeps its compare: in
    both integrators the env whose velocity is set to inf is done after THIS step, as in the reference."""
    n = 256
    env = scg.make('quadrotor', quad_type=3, num_envs=n, device='cuda', seed=3, randomized_init=True, task='traj_tracking',
                   cost='quadratic', physics=physics, info_level='minimal')
    env.reset()
    hover = torch.as_tensor(np.tile(np.asarray(env.U_GOAL, dtype=np.float32), (n, 1)), device='cuda')
    env.state[3, 1] = float('inf')           # x velocity
    env.state[4, 11] = float('-inf')         # yaw rate
    env.state[5, 5] = float('nan')           # z velocity
    obs, rew, done, _ = env.step(hover)
    assert bool(done[3]) and bool(done[4]) and bool(done[5])
    assert int(done.sum()) == 3 and torch.isfinite(obs).all()
    env.close()


def test_rollout_is_capturable_in_a_hip_graph(gpu):
    """env.rollout() with one launch queue only enqueues kernels on the current stream, so it can be captured in a HIP graph
    (torch.cuda.graph) and replayed: the replay leaves the same buffers as the plain call.  The reset serials are kernel
    arguments, i.e. part of the capture: every replay of one graph draws the same reset offsets (documented in INTEGRATION.md)."""
    n, K = 1 << 14, 40
    kw = dict(num_envs=n, device='cuda', seed=9, randomized_init=True, task='stabilization', cost='quadratic', ctrl_freq=50,
              pyb_freq=50, episode_len_sec=0.3, info_level='minimal', track_episode_stats=True, rollout_queues=1)
    a, b = scg.make('cartpole', **kw), scg.make('cartpole', **kw)
    a.reset(), b.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    lo, hi = float(a.action_space.low[0]), float(a.action_space.high[0])
    acts = torch.rand((7, 1, a.plane_stride), generator=g, device='cuda') * (hi - lo) + lo
    a.rollout(acts, steps=10)
    a.rollout(acts, steps=K)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        b.rollout(acts, steps=10)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            b.rollout(acts, steps=K)
    torch.cuda.synchronize()
    assert int(b.ctrl_step_counter.max()) <= 10                       # the capture itself ran nothing
    graph.replay()
    torch.cuda.synchronize()
    for key in ('state', 'step_count', 'ep_return', 'reward', 'done'):
        assert torch.equal(a._bufs[key], b._bufs[key]), key
    a.close(), b.close()


@pytest.mark.parametrize('system', ['cartpole', 2, 3, 'noisy'])
def test_rollout_graph_replay_is_bit_identical_to_plain_launches(gpu, system):
    """Small batches: scg_rollout captures the launches of a signature it sees for the second time into a hipGraph and replays it
    (scg_config.rollout_graph).  The step serial travels through a device word, so the Philox reset draws are those of plain
    launches: state, counters, returns, reward, done equal a twin env with rollout_graph=-1 bit for bit after every call --
    first sighting (plain), capture + replay, replays, a second signature, a reset in between, a changed action tensor."""
    n = 5000
    kw = dict(num_envs=n, device='cuda', seed=21, randomized_init=True, ctrl_freq=50, pyb_freq=50, episode_len_sec=0.3,
              info_level='minimal', track_episode_stats=True)
    if system == 'cartpole':
        mk = lambda **e: scg.make('cartpole', task='stabilization', cost='quadratic', **kw, **e)
    elif system == 'noisy':
        # the disturbance tier draws its noise from Philox streams keyed by the step serial BEFORE the state advance: the serial
        # word of a replayed launch must be there in time (plus a per-episode impulse offset: episode ids = reset serials)
        dist = {'action': [{'disturbance_func': 'white_noise', 'std': 0.002}],
                'observation': [{'disturbance_func': 'white_noise', 'std': 0.01}],
                'dynamics': [{'disturbance_func': 'impulse', 'magnitude': 0.01, 'step_offset': None, 'duration': 3, 'decay_rate': 0.8}]}
        mk = lambda **e: scg.make('quadrotor', quad_type=2, task='stabilization', cost='rl_reward', disturbances=dist, **kw, **e)
    else:
        mk = lambda **e: scg.make('quadrotor', quad_type=system, task='traj_tracking', cost='rl_reward', **kw, **e)
    plain, graphed = mk(rollout_graph=-1), mk()
    plain.reset(), graphed.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(4)
    lo = torch.as_tensor(plain.action_space.low, device='cuda')[None, :, None]
    hi = torch.as_tensor(plain.action_space.high, device='cuda')[None, :, None]
    acts = torch.rand((9, plain.action_dim, plain.plane_stride), generator=g, device='cuda') * (hi - lo) + lo
    acts2 = acts.flip(0).contiguous()

    def same(tag):
        torch.cuda.synchronize()
        for key in ('state', 'step_count', 'ep_return', 'reward', 'done', 'obs', 'episode_id'):
            if key in plain._bufs:
                assert torch.equal(plain._bufs[key], graphed._bufs[key]), (tag, key)

    for call in range(4):                                  # plain, capture + replay, replay, replay
        plain.rollout(acts, steps=24), graphed.rollout(acts, steps=24)
        same('call %d' % call)
    cached, replays = graphed._engine.rollout_graph_info()
    assert cached == 1 and replays == 3 and plain._engine.rollout_graph_info() == (0, 0)
    for call in range(3):                                  # a second signature (steps), the first stays cached
        plain.rollout(acts, steps=10), graphed.rollout(acts, steps=10)
        same('steps 10, call %d' % call)
    assert graphed._engine.rollout_graph_info() == (2, 5)
    plain.reset(), graphed.reset()                         # reset() moves the serial; the cached graph picks the new one up
    plain.rollout(acts, steps=24), graphed.rollout(acts, steps=24)
    same('after reset')
    assert graphed._engine.rollout_graph_info() == (2, 6)
    for call in range(2):                                  # another action tensor is another signature
        plain.rollout(acts2, steps=24), graphed.rollout(acts2, steps=24)
        same('other actions, call %d' % call)
    assert graphed._engine.rollout_graph_info() == (3, 7)
    assert int(graphed._bufs['done'].sum()) > 0 or int(graphed.ctrl_step_counter.min()) < 24      # auto-resets happened on the way
    plain.close(), graphed.close()


def test_state_out_through_the_c_abi(gpu):
    """scg_step_io.state_out: the new state goes to a buffer of the caller's, `state` is only read; scg_rollout refuses it."""
    from safe_control_gym_amd import _capi
    n = 1000
    env = scg.make('quadrotor', quad_type=3, num_envs=n, device='cuda', seed=4, randomized_init=True, info_level='minimal',
                   task='traj_tracking', cost='quadratic')
    env.reset()
    eng, b, ld = env._engine, env._bufs, env.plane_stride
    before = b['state'].clone()
    a = torch.zeros((4, ld), device='cuda'); a[:, :n] = torch.as_tensor(np.asarray(env.U_GOAL, dtype=np.float32), device='cuda')[:, None] * 1.1
    out = torch.full((12, ld), float('nan'), device='cuda')
    io = dict(b); io['action'] = a; io['state_out'] = out
    eng.step(io, n, ld, env._serial)
    torch.cuda.synchronize()
    assert torch.equal(b['state'], before)                               # the input was only read
    step_count = b['step_count'].clone()
    b['step_count'].zero_()
    if 'ep_return' in b:
        b['ep_return'].zero_()
    io2 = dict(b); io2['action'] = a
    eng.step(io2, n, ld, env._serial)                                    # the same step in place
    assert torch.equal(out[:, :n], b['state'][:, :n]) and torch.isnan(out[:, n:]).all()
    assert torch.equal(step_count[:n], b['step_count'][:n])
    acts = a[None].contiguous()
    with pytest.raises(_capi.ScgError, match='state_out'):
        eng.rollout(dict(io), n, ld, env._serial, 2, acts)
    env.close()


def test_reference_push_after_step_protocol_keeps_the_previous_observation(gpu):
    """The reference's rollout loop (controllers/ppo/ppo.py:259-276) steps the env and runs the observation normaliser
    BEFORE it pushes the previous ``obs`` into the buffer.  With a plain ``make()`` env (copy_outputs defaults to True
    with info_level='full') and the normaliser's fresh-tensor default that order stores obs[t] under obs[t], not
    obs[t+1]; the zero-copy views of copy_outputs=False / ``out=`` are an opt-in that really alias."""
    from safe_control_gym_amd.math_and_models.normalization import MeanStdNormalizer
    n, T = 4096, 6
    env = scg.make('quadrotor', quad_type=2, num_envs=n, device='cuda', seed=3, randomized_init=True)
    norm = MeanStdNormalizer(shape=(env.obs_dim,))
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    lo, hi = float(env.action_space.low[0]), float(env.action_space.high[0])
    stored, truth = [], []
    raw = env.reset()
    obs = norm(raw)
    for t in range(T):
        truth.append(obs.clone())                                   # what obs[t] really is, copied at time t
        act = torch.rand((n, 2), generator=g, device='cuda') * (hi - lo) + lo
        next_raw, rew, done, info = env.step(act)
        next_obs = norm(next_raw)
        stored.append(obs)                                          # push AFTER step + normaliser, as ppo.py does (no clone)
        obs = next_obs
    torch.cuda.synchronize()
    for t in range(T):
        assert torch.equal(stored[t], truth[t]), t
        if t + 1 < T:
            assert not torch.equal(stored[t], stored[t + 1])
    assert raw.data_ptr() != env._bufs['state'].data_ptr()          # reset() handed out a fresh tensor as well
    env.close()
    # the opt-ins alias, as documented
    env = scg.make('quadrotor', quad_type=2, num_envs=n, device='cuda', seed=3, randomized_init=True, copy_outputs=False)
    o0 = env.reset()
    act = torch.rand((n, 2), generator=g, device='cuda') * (hi - lo) + lo
    o1 = env.step(act)[0]
    assert o0.data_ptr() == o1.data_ptr() == env._bufs['state'].data_ptr() and torch.equal(o0, o1)
    out = torch.empty((env.obs_dim, (n + 63) // 64 * 64), dtype=torch.float32, device='cuda')
    y0 = norm(o1, out=out)
    y1 = norm(env.step(act)[0], out=out)
    assert y0.data_ptr() == y1.data_ptr() == out.data_ptr()
    assert scg.make('quadrotor', quad_type=2, num_envs=8, device='cuda', info_level='minimal').COPY_OUTPUTS is False
    env.close()


def test_reseed_on_reset_repeats_the_draws_and_keeps_the_handle(gpu):
    """benchmark_env.py:356-362: with reseed_on_reset every reset() re-seeds first, so consecutive resets return the SAME
    randomised initial states (and the disturbance streams restart); without it they differ.  The Philox key is unchanged,
    so the scg_handle is kept (no hipFree / hipMalloc of the tables per reset)."""
    n = 3000
    kw = dict(quad_type=3, num_envs=n, device='cuda', seed=11, randomized_init=True, randomized_inertial_prop=True,
              task='traj_tracking', disturbances={'observation': [{'disturbance_func': 'white_noise', 'std': 0.01}]})
    env = scg.make('quadrotor', reseed_on_reset=True, **kw)
    a = env.reset().clone()
    pa = env.inertial_properties.clone()
    handle = env._engine
    act = torch.zeros((n, 4), device='cuda') + float(env.action_space.low[0])
    s1 = [tuple(t.clone() for t in env.step(act)[:3]) for _ in range(3)]
    b = env.reset().clone()
    assert env._engine is handle and env._engine.handle
    assert torch.equal(a, b) and torch.equal(pa, env.inertial_properties)
    s2 = [tuple(t.clone() for t in env.step(act)[:3]) for _ in range(3)]
    for (o1, r1, d1), (o2, r2, d2) in zip(s1, s2):                  # the in-kernel noise streams restart with the reset
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    env.close()
    env = scg.make('quadrotor', **kw)
    a = env.reset().clone()
    b = env.reset().clone()
    assert not torch.equal(a, b)
    env.seed(11)                                                    # an explicit re-seed rewinds the stream too
    assert torch.equal(env.reset(), a)
    env.close()


def test_checkpoint_carries_the_separate_observation_buffer(gpu):
    """An env whose observation is a buffer of its own (goal horizon + observation noise) exposes the SAME observation
    after load_state_dict as the env the snapshot was taken from, before any further step."""
    n = 777
    kw = dict(quad_type=2, num_envs=n, device='cuda', seed=2, task='traj_tracking', cost='rl_reward', obs_goal_horizon=2,
              randomized_init=True, disturbances={'observation': [{'disturbance_func': 'white_noise', 'std': 0.05}]})
    a = scg.make('quadrotor', **kw)
    a.reset()
    act = torch.zeros((n, 2), device='cuda') + float(a.action_space.high[0])
    for _ in range(7):
        obs_a = a.step(act)[0]
    b = scg.make('quadrotor', **kw)
    b.load_state_dict(a.state_dict())
    assert 'obs' in a.state_dict()['planes']
    assert torch.equal(b._obs_view(), obs_a) and not torch.equal(b._obs_view()[:, :6], b.state)   # noise is in obs only
    oa, ob = a.step(act)[0], b.step(act)[0]
    assert torch.equal(oa, ob)
    a.close(), b.close()


def test_two_handles_in_one_process_may_differ_in_launch_policy(gpu):
    """The launch policy is per handle (scg_config.rollout_queues / ref_in_lds / grid_cap_per_cu), not per process: three
    envs with different policies coexist and produce bit-identical buffers."""
    n = 300001
    kw = dict(quad_type=3, num_envs=n, device='cuda', seed=5, randomized_init=True, task='traj_tracking', cost='quadratic',
              episode_len_sec=0.25, info_level='minimal')
    envs = [scg.make('quadrotor', rollout_split_min=131072, **kw), scg.make('quadrotor', rollout_queues=1, **kw),
            scg.make('quadrotor', rollout_queues=2, rollout_split_min=1024, ref_in_lds=1, **kw),
            scg.make('quadrotor', rollout_queues=1, grid_cap_per_cu=2, **kw)]
    g = torch.Generator(device='cuda'); g.manual_seed(8)
    lo, hi = float(envs[0].action_space.low[0]), float(envs[0].action_space.high[0])
    acts = torch.rand((5, 4, envs[0].plane_stride), generator=g, device='cuda') * (hi - lo) + lo
    for e in envs:
        e.reset()
        e.rollout(acts, steps=17)
    torch.cuda.synchronize()
    assert envs[2]._engine.kernel_info(n)['lds_bytes'] > envs[0]._engine.kernel_info(n)['lds_bytes']
    assert envs[3]._engine.kernel_info(n)['grid'] < envs[0]._engine.kernel_info(n)['grid']
    for e in envs[1:]:
        for key in ('state', 'step_count', 'reward', 'done', 'ep_return'):
            assert torch.equal(envs[0]._bufs[key], e._bufs[key]), key
    for e in envs:
        e.close()


def test_all_gather_of_episode_returns_through_rccl(gpu):
    """The nccl (= RCCL) branch of parallel.all_gather_episode_returns on this one-GPU box: a world of one rank, the
    collective forced (a one-rank world otherwise returns a copy without touching the backend)."""
    import torch.distributed as dist
    from safe_control_gym_amd import parallel
    if dist.is_initialized():
        dist.destroy_process_group()
    import tempfile
    with tempfile.NamedTemporaryFile() as f:
        dist.init_process_group(backend='nccl', init_method='file://' + f.name + '.store', rank=0, world_size=1)
        try:
            assert dist.get_backend() == 'nccl'
            x = torch.arange(1000, dtype=torch.float32, device='cuda') * 0.25
            y = parallel.all_gather_episode_returns(x, total_envs=1000, force_collective=True)
            z = parallel.all_gather_episode_returns(x, force_collective=True)      # sizes discovered by a collective
            ones = torch.ones(1, device='cuda')
            dist.all_reduce(ones)
            torch.cuda.synchronize()
            assert torch.equal(x, y) and torch.equal(x, z) and y.data_ptr() != x.data_ptr() and int(ones.item()) == 1
            # the asynchronous form bench.py uses: RCCL gathers a snapshot on its own stream while step kernels keep updating
            # the plane; the gathered values are those at the time of the call
            env = scg.make('quadrotor', quad_type=3, num_envs=1 << 16, device='cuda', seed=3, randomized_init=True,
                           task='traj_tracking', cost='quadratic', info_level='minimal', track_episode_stats=True)
            env.reset()
            acts = torch.zeros((1, 4, env.plane_stride), device='cuda') + float(env.action_space.high[0]) * 0.6
            env.rollout(acts, steps=7)
            before = env.episode_return.clone()
            pending = parallel.begin_all_gather_episode_returns(env.episode_return, total_envs=1 << 16, force_collective=True)
            assert pending._work is not None                           # the RCCL path, not the deferred one
            env.rollout(acts, steps=30)
            got = pending.wait()
            torch.cuda.synchronize()
            assert torch.equal(got, before) and not torch.equal(env.episode_return, before)
            # the normalisers' shared statistics (sync_statistics: moments -> all-reduce of the sums -> update + normalise) with the
            # all-reduce on RCCL, device doubles in place; a world of one rank must reproduce the plain normalisers
            from safe_control_gym_amd.math_and_models.normalization import MeanStdNormalizer, RewardStdNormalizer
            obs, rew, done, _ = env.step(env.action_buffer)
            plain, shared = MeanStdNormalizer(shape=(env.obs_dim,)), MeanStdNormalizer(shape=(env.obs_dim,), sync_statistics='force')
            rp, rs = RewardStdNormalizer(), RewardStdNormalizer(sync_statistics='force')
            for _ in range(3):
                a, b = plain(obs), shared(obs)
                ra, rb = rp(rew, done), rs(rew, done)
                assert float((a - b).abs().max()) <= 1e-6 and float((ra - rb).abs().max()) <= 1e-6
                obs, rew, done, _ = env.step(env.action_buffer)
            assert torch.allclose(plain.rms.mean, shared.rms.mean, rtol=1e-12, atol=1e-12) and torch.allclose(plain.rms.var, shared.rms.var, rtol=1e-9)
            assert torch.allclose(rp.rms.var, rs.rms.var, rtol=1e-9) and torch.equal(rp.ret, rs.ret)
            env.close()
        finally:
            dist.destroy_process_group()


def test_bench_multi_gpu_code_path_on_rccl_with_one_rank(gpu):
    """SCG_BENCH_FORCE_COLLECTIVES=1: bench.py builds a ONE-rank RCCL world and runs every collective of its N > 1 path on the
    device -- the barriers of the timed regions, the max-reduce of the times, the asynchronous all-gather of episode returns
    inside the timed region, the gather of per-rank step times.  (RCCL refuses two ranks on one device, so this is as close as
    a one-GPU box gets to the sharded run.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['SCG_BENCH_FORCE_COLLECTIVES'] = '1'
    run = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '320', '--warmup', '5', '--repeats', '2',
                          '--envs-per-gpu', '65536', '--no-cpu-baseline', '--no-sub-records'], env=env, capture_output=True, text=True,
                         timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    j = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith('{')][-1])
    assert j['backend'] == 'nccl' and j['ranks_seen'] == 1 and j['forced_collectives'] is True and j['value'] > 0
    assert len(j['per_rank_ms_per_step_min_max']) == 2


def test_bench_launcher_runs_two_ranks_on_this_gpu(gpu):
    """`python bench.py --gpus 2` with no launcher environment: the parent (no GPU call) starts two rank processes, which here share
    the box's one GPU and exchange over gloo (SCG_SINGLE_DEVICE=1: RCCL refuses two ranks on one device); rank 0's line comes back
    through the parent with ranks_seen = 2, the sharded workload C5 and a positive throughput.  3 processes touch the card (<= 6)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['SCG_SINGLE_DEVICE'] = '1'
    run = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '5',
                          '--envs-per-gpu', '65536', '--repeats', '2'], env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['ranks_seen'] == 2 and j['backend'] == 'gloo' and j['steps'] == 20
    assert j['config']['total_envs'] == 131072 and j['config']['workload'].startswith('C5')
    # (no throughput bar: two ranks time-share one GPU and exchange through host memory)
    assert j['value'] > 0 and 'per_rank_ms_per_step_min_max' in j and 'cpu_baseline' not in j
