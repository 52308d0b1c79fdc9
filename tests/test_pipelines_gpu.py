"""Round 4: the step as the reference's VecEnv protocol splits it (step_async / step_wait on per-shard launch queues,
/root/reference/safe_control_gym/envs/env_wrappers/vectorized_env/vec_env.py:47-72, subproc_vec_env.py:50-63) and the
trajectory sink of the multi-step rollouts (what tests/scripts/verbose_api.py:71-108 prints per step and
controllers/ppo/ppo_utils.py:308-336 stores).  Everything here is an EQUALITY test: a pipelined loop, a sunk rollout and a
ring of output buffers must reproduce the plain ``step()`` loop bit for bit (the oracle parity of that loop is
tests/test_parity_gpu.py's business)."""
from functools import partial

import numpy as np
import pytest
import torch

import safe_control_gym_amd as scg
from safe_control_gym_amd.envs.env_wrappers.vectorized_env import make_vec_envs, pipeline_bounds

pytestmark = pytest.mark.gpu

C3_CONSTRAINTS = [{'constraint_form': 'bounded_constraint', 'constrained_variable': 'state', 'active_dims': [0, 2, 4],
                   'lower_bounds': [-2, 0, -0.5], 'upper_bounds': [2, 2, 0.5]}]

TIERS = {
    # the lean tier of bench.py (zero-copy views), C4
    'lean_views': ('quadrotor', dict(quad_type=3, task='traj_tracking', cost='quadratic', info_level='minimal')),
    # a plain make(): full info, fresh outputs by ping-pong state
    'default_fresh': ('quadrotor', dict(quad_type=3, task='traj_tracking', cost='quadratic')),
    # the usual RL configuration: goal horizon appended to the observation, fresh outputs, constraints with a penalty
    'goal_horizon_fresh': ('quadrotor', dict(quad_type=2, task='traj_tracking', cost='rl_reward', obs_goal_horizon=2,
                                             constraints=C3_CONSTRAINTS, use_constraint_penalty=True, normalized_rl_action_space=True)),
    # an observation disturbance: kernel-written obs buffer of its own, in-kernel Philox noise keyed by the global env index
    'obs_noise': ('quadrotor', dict(quad_type=2, task='stabilization', cost='rl_reward',
                                    disturbances={'observation': [{'disturbance_func': 'white_noise', 'std': 0.01}]})),
    'cartpole_views': ('cartpole', dict(task='stabilization', cost='quadratic', info_level='minimal', ctrl_freq=50, pyb_freq=50)),
}


def _common(kw):
    c = dict(device='cuda', seed=11, ctrl_freq=60, pyb_freq=240, episode_len_sec=1, randomized_init=True)
    c.update(kw)
    return c


def _policy(env):
    """A device-side stand-in policy with a real dependency on the observation: an affine map of the first nu observed
    states, squashed into the action space (row-major [n, nu] float32, what a torch policy emits)."""
    nu = env.action_dim
    lo = torch.as_tensor(env.action_space.low, device='cuda', dtype=torch.float32)
    hi = torch.as_tensor(env.action_space.high, device='cuda', dtype=torch.float32)
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)

    def policy(obs, k):
        z = torch.tanh(3.0 * obs[:, :nu] + 0.37 * k)
        return (mid + 0.9 * half * z).contiguous()
    return policy


def _same(a, b, what):
    if isinstance(a, dict):
        for key in a:
            if key != 'n':
                _same(a[key], b[key], '%s[%r]' % (what, key))
        return
    assert a.shape == b.shape, what
    eq = (a == b) | (torch.isnan(a) & torch.isnan(b)) if a.is_floating_point() else (a == b)
    assert bool(eq.all()), '%s differs in %d of %d elements' % (what, int((~eq).sum()), eq.numel())


@pytest.mark.parametrize('tier', list(TIERS))
@pytest.mark.parametrize('on_queue', [False, True], ids=['caller_stream', 'policy_on_the_shard_queue'])
def test_pipelined_step_loop_equals_the_plain_step_loop(gpu, tier, on_queue):
    """make_vec_envs(num_pipelines=2): per-shard step_async / step_wait with a policy in the loop == env.step() on the
    unsplit batch, bit for bit -- observations, rewards, dones and every info tensor, through auto-resets and time limits."""
    name, kw = TIERS[tier]
    n, steps = 3000, 90                              # episodes of 60 (50 for the cart-pole) steps: every env hits its time limit once
    plain = scg.make(name, num_envs=n, **_common(kw))
    venv = make_vec_envs(partial(scg.make, name, **_common(kw)), batch_size=n, num_pipelines=2)
    assert venv.num_pipelines == 2 and venv.bounds == [(0, 1280), (1280, 3000)] == pipeline_bounds(n, 2)
    policy = _policy(plain)
    obs = plain.reset()
    pobs = [venv.reset(pipeline=p)[0] for p in range(2)]
    _same(torch.cat(pobs), obs, 'reset obs')
    streams = [venv.pipeline_stream(p) if on_queue else torch.cuda.current_stream() for p in range(2)]
    torch.cuda.synchronize()
    hits = 0
    for k in range(steps):
        obs, rew, done, info = plain.step(policy(obs, k))
        outs = []
        for p in range(2):
            with torch.cuda.stream(streams[p]):
                venv.step_async(policy(pobs[p], k), pipeline=p)
        for p in range(2):
            with torch.cuda.stream(streams[p]):
                outs.append(venv.step_wait(pipeline=p))
        if on_queue:
            for s in streams:
                torch.cuda.current_stream().wait_stream(s)
        pobs = [o[0] for o in outs]
        _same(torch.cat(pobs), obs, 'obs at step %d' % k)
        _same(torch.cat([o[1] for o in outs]), rew, 'reward at step %d' % k)
        _same(torch.cat([o[2] for o in outs]), done, 'done at step %d' % k)
        merged = {key: (torch.cat([o[3][key] for o in outs]) if not isinstance(outs[0][3][key], dict) else
                        {q: torch.cat([o[3][key][q] for o in outs]) for q in outs[0][3][key]}) for key in outs[0][3] if key != 'n'}
        if 'terminal_observation' in merged:          # records are written for finishing envs only: compare those rows
            d = done.bool()
            for key in ('terminal_observation',):
                _same(merged[key][d], info[key][d], '%s at step %d' % (key, k))
            if 'episode' in merged:
                _same(merged['episode']['r'][d], info['episode']['r'][d], 'episode r at step %d' % k)
                _same(merged['episode']['l'][d], info['episode']['l'][d], 'episode l at step %d' % k)
            merged = {q: v for q, v in merged.items() if q not in ('terminal_observation', 'episode')}
            info = {q: v for q, v in info.items() if q not in ('terminal_observation', 'episode', 'n')}
        _same(merged, {q: v for q, v in info.items() if q != 'n'}, 'info at step %d' % k)
        hits += int(done.sum())
    assert hits >= n                                   # every env finished an episode inside the comparison
    # the whole-batch calls of the protocol (actions split by rows, outputs concatenated)
    a = policy(obs, steps)
    o1 = plain.step(a)
    o2 = venv.step(a)
    for q, what in enumerate(('obs', 'reward', 'done')):
        _same(o2[q], o1[q], 'whole-batch ' + what)
    venv.close(); plain.close()


def test_step_async_protocol_errors(gpu):
    env = scg.make('quadrotor', quad_type=2, num_envs=256, **_common({}))
    obs = env.reset()
    a = torch.as_tensor(env.action_space.sample(256), device='cuda', dtype=torch.float32)
    with pytest.raises(RuntimeError, match='without a pending'):
        env.step_wait()
    env.step_async(a)
    with pytest.raises(RuntimeError, match='pending'):
        env.step(a)
    with pytest.raises(RuntimeError, match='pending'):
        env.step_async(a)
    with pytest.raises(RuntimeError, match='pending'):
        env.rollout(torch.zeros((1, 2, env.plane_stride), device='cuda'))
    for call in (env.reset, env.state_dict, lambda: env.load_state_dict({})):
        with pytest.raises(RuntimeError, match='pending'):
            call()
    o, r, d, info = env.step_wait()
    twin = scg.make('quadrotor', quad_type=2, num_envs=256, **_common({}))
    twin.reset()
    o2, r2, d2, _ = twin.step(a)
    assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(d, d2)
    # a synchronous step behind an asynchronous one is ordered behind it by the library
    env.step_async(a); env.step_wait()
    o, r, d, _ = env.step(a)
    twin.step(a)
    o2, r2, d2, _ = twin.step(a)
    assert torch.equal(o, o2) and torch.equal(r, r2)
    env.close(); twin.close()


@pytest.mark.parametrize('mode', ['action', 'dynamics'])
def test_asynchronous_step_with_an_adversary_equals_the_plain_step(gpu, mode):
    """benchmark_env.py:256-266: the adversary's planes are built per step and read by the env's own queue; between
    step_async and step_wait the caller allocates freely (the planes must stay alive until the join)."""
    n, steps = 2048, 40
    kw = _common(dict(quad_type=2, task='stabilization', cost='rl_reward', adversary_disturbance=mode, adversary_disturbance_scale=0.02))
    a_env, b_env = scg.make('quadrotor', num_envs=n, **kw), scg.make('quadrotor', num_envs=n, **kw)
    policy = _policy(a_env)
    oa, ob = a_env.reset(), b_env.reset()
    g = torch.Generator(device='cuda').manual_seed(5)
    adv_dim = a_env.DISTURBANCE_MODES[mode]['dim']
    for k in range(steps):
        adv = torch.rand((n, adv_dim), device='cuda', generator=g) * 2 - 1
        a_env.set_adversary_control(adv); b_env.set_adversary_control(adv)
        oa, ra, da, _ = a_env.step(policy(oa, k))
        b_env.step_async(policy(ob, k))
        churn = [torch.full((3, b_env.plane_stride), float(k), device='cuda') for _ in range(4)]     # reuses freed blocks at once
        del churn
        ob, rb, db, _ = b_env.step_wait()
        _same(ob, oa, 'obs at step %d' % k); _same(rb, ra, 'reward at step %d' % k); _same(db, da, 'done at step %d' % k)
    a_env.close(); b_env.close()


PLANE_TIERS = {
    'info': dict(quad_type=3, task='traj_tracking', cost='quadratic'),
    'info_constraints_2d': dict(quad_type=2, task='stabilization', cost='rl_reward', constraints=C3_CONSTRAINTS),
    'goal_horizon': dict(quad_type=2, task='traj_tracking', cost='rl_reward', obs_goal_horizon=2),
    'stabilisation_goal_in_obs': dict(quad_type=2, task='stabilization', cost='rl_reward', obs_goal_horizon=1),
    'obs_noise': dict(quad_type=2, task='stabilization', cost='rl_reward', disturbances={'observation': [{'disturbance_func': 'white_noise', 'std': 0.01}]}),
    'inertial': dict(quad_type=3, task='traj_tracking', cost='quadratic', randomized_inertial_prop=True),
    'catch_all_drag': dict(quad_type=3, task='traj_tracking', cost='quadratic', physics='pyb_drag'),
    'cartpole': dict(task='stabilization', cost='quadratic', ctrl_freq=50, pyb_freq=50),
}


@pytest.mark.parametrize('tier', list(PLANE_TIERS))
def test_terminal_planes_hold_what_the_terminal_records_hold(gpu, tier):
    """terminal_layout='planes' (scg_step_io.terminal_stride == 0: SoA planes [obs_dim + 2][ld] instead of one dense record per env):
    for every env that finishes, info['terminal_observation'] / info['episode'] equal the record layout's, everything else equals bit
    for bit; through step() and through a pipelined shard pair (the shard's planes are the whole batch's, shifted)."""
    kw = _common(PLANE_TIERS[tier])
    name = 'cartpole' if tier == 'cartpole' else 'quadrotor'
    n, steps = 2500, 75
    a_env = scg.make(name, num_envs=n, **kw)
    b_env = scg.make(name, num_envs=n, terminal_layout='planes', **kw)
    venv = make_vec_envs(partial(scg.make, name, terminal_layout='planes', **kw), batch_size=n, num_pipelines=2)
    policy = _policy(a_env)
    oa, ob = a_env.reset(), b_env.reset()
    assert b_env._bufs['terminal_record'].shape == (b_env.obs_dim + 2, b_env.plane_stride)
    oc = [venv.reset(pipeline=p)[0] for p in range(2)]
    hits = 0
    for k in range(steps):
        act = policy(oa, k)
        oa, ra, da, ia = a_env.step(act)
        ob, rb, db, ib = b_env.step(act)
        for p, (lo, hi) in enumerate(venv.bounds):
            venv.step_async(act[lo:hi], pipeline=p)
        outs = [venv.step_wait(pipeline=p) for p in range(2)]
        _same(ob, oa, 'obs'); _same(rb, ra, 'reward'); _same(db, da, 'done')
        _same(torch.cat([o[0] for o in outs]), oa, 'shard obs')
        d = da.bool()
        if bool(d.any()):
            hits += int(d.sum())
            _same(ib['terminal_observation'][d], ia['terminal_observation'][d], 'terminal observation at step %d' % k)
            _same(torch.cat([o[3]['terminal_observation'] for o in outs])[d], ia['terminal_observation'][d], 'shard terminal observation')
            if 'episode' in ia:
                _same(ib['episode']['r'][d], ia['episode']['r'][d], 'episode return'); _same(ib['episode']['l'][d], ia['episode']['l'][d], 'episode length')
                _same(torch.cat([o[3]['episode']['l'] for o in outs])[d], ia['episode']['l'][d], 'shard episode length')
        _same(ib['TimeLimit.truncated'], ia['TimeLimit.truncated'], 'truncated')
    assert hits >= n
    a_env.close(); b_env.close(); venv.close()


SINK_CASES = {
    'c4_lean': ('quadrotor', dict(quad_type=3, task='traj_tracking', cost='quadratic', info_level='minimal')),
    'c3_constraints': ('quadrotor', dict(quad_type=2, task='stabilization', cost='rl_reward', constraints=C3_CONSTRAINTS,
                                         info_level='minimal')),
    'c2_cartpole': ('cartpole', dict(task='stabilization', cost='quadratic', info_level='minimal', ctrl_freq=50, pyb_freq=50)),
    'goal_horizon_info': ('quadrotor', dict(quad_type=2, task='traj_tracking', cost='rl_reward', obs_goal_horizon=1, copy_outputs=False)),
    'obs_noise': ('quadrotor', dict(quad_type=1, task='stabilization', cost='rl_reward', copy_outputs=False,
                                    disturbances={'observation': [{'disturbance_func': 'white_noise', 'std': 0.01}]})),
}


@pytest.mark.parametrize('case', list(SINK_CASES))
@pytest.mark.parametrize('form', ['launches', 'two_queues', 'fused'])
def test_rollout_trajectory_equals_the_step_loop(gpu, case, form):
    """rollout(actions, out=trajectory): row k of the buffer == what step() returned at step k (observation incl. the reset
    observation of finished envs, reward, done, constraint values, mse, truncated), for per-step launches on one and on two
    queues and for the one-launch fused form; the env afterwards continues exactly like its step()-driven twin."""
    name, kw = SINK_CASES[case]
    n, steps = 2500, 75
    extra = dict(rollout_split_min=256) if form == 'two_queues' else dict(rollout_queues=1)
    a_env = scg.make(name, num_envs=n, **_common(kw))
    b_env = scg.make(name, num_envs=n, **_common(dict(kw, **extra)))
    a_env.reset(); b_env.reset()
    ld, nu = a_env.plane_stride, a_env.action_dim
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    lo = torch.as_tensor(a_env.action_space.low, device='cuda')[None, :, None]
    hi = torch.as_tensor(a_env.action_space.high, device='cuda')[None, :, None]
    acts = torch.rand((9, nu, ld), generator=g, device='cuda') * (hi - lo) * 1.1 + lo - 0.05 * (hi - lo)
    info_sink = a_env.INFO_LEVEL == 'full'
    traj = b_env.trajectory_buffer(steps, info=info_sink)
    assert b_env.rollout(acts, steps=steps, fused=(form == 'fused'), out=traj) is traj and traj.filled == steps
    finished = 0
    for k in range(steps):
        obs, rew, done, info = a_env.step(acts[k % 9])
        _same(traj.obs[k], obs, 'obs at step %d' % k)
        _same(traj.reward[k], rew, 'reward at step %d' % k)
        _same(traj.done[k], done, 'done at step %d' % k)
        if a_env.constraints is not None:
            _same(traj.constraint_values[k], info['constraint_values'], 'constraint values at step %d' % k)
            _same(traj.constraint_violation[k], info['constraint_violation'].bool(), 'violation at step %d' % k)
        if info_sink:
            _same(traj.mse[k], info['mse'], 'mse at step %d' % k)
            _same(traj.truncated[k], info['TimeLimit.truncated'], 'truncated at step %d' % k)
        finished += int(done.sum())
    assert finished >= n
    assert torch.equal(a_env.state, b_env.state) and torch.equal(a_env.ctrl_step_counter, b_env.ctrl_step_counter)
    assert torch.equal(a_env.episode_return, b_env.episode_return)
    # ... and both continue alike (the state came home from the sink)
    o1 = a_env.step(acts[3]); o2 = b_env.step(acts[3])
    assert torch.equal(o1[0], o2[0]) and torch.equal(o1[1], o2[1]) and torch.equal(o1[2], o2[2])
    # a second rollout into the same buffer, shorter: rows beyond `filled` are stale, the first ones are the new steps
    b_env.rollout(acts, steps=5, fused=(form == 'fused'), out=traj)
    for k in range(5):
        obs, rew, done, _ = a_env.step(acts[k % 9])
        _same(traj.obs[k], obs, 'second rollout: obs at step %d' % k)
        _same(traj.reward[k], rew, 'second rollout: reward at step %d' % k)
    assert traj.filled == 5
    a_env.close(); b_env.close()


def test_trajectory_buffer_argument_checks(gpu):
    env = scg.make('quadrotor', quad_type=2, num_envs=300, **_common(dict(info_level='minimal')))
    other = scg.make('quadrotor', quad_type=2, num_envs=301, **_common(dict(info_level='minimal')))
    env.reset(); other.reset()
    acts = torch.zeros((4, 2, env.plane_stride), device='cuda')
    with pytest.raises(ValueError, match='another env or fewer steps'):
        env.rollout(acts, steps=4, out=other.trajectory_buffer(4))
    with pytest.raises(ValueError, match='another env or fewer steps'):
        env.rollout(acts, steps=4, out=env.trajectory_buffer(3))
    with pytest.raises(ValueError, match='no constraints'):
        env.trajectory_buffer(4, constraints=True)
    with pytest.raises(ValueError, match='device tensor'):          # a host tensor's address must never reach a kernel
        env.rollout(acts.cpu(), steps=4)
    env.close(); other.close()


@pytest.mark.parametrize('n, extra', [
    (1000, {}),                                     # state ping-pong, plane stride != n
    (1024, {}),                                     # ... plane stride == n (the views are plain transposes)
    (1000, dict(obs_goal_horizon=1)),               # goal rows behind the state: the slot's state rows are a view of its planes
    (1024, dict(disturbances={'observation': [{'disturbance_func': 'white_noise', 'std': 0.01}]})),   # a kernel-written obs buffer
])
def test_output_ring_reproduces_the_fresh_output_loop(gpu, n, extra):
    """output_ring=K: step() cycles through K pre-allocated output sets (whose views are made once) instead of allocating one per
    step -- the same numbers; a tensor handed out is overwritten K steps later (the documented, opt-in change of ownership)."""
    kw = _common(dict(quad_type=3, task='traj_tracking', cost='quadratic', **extra))
    a_env = scg.make('quadrotor', num_envs=n, **kw)
    b_env = scg.make('quadrotor', num_envs=n, output_ring=3, **kw)
    with pytest.raises(ValueError, match='output_ring'):
        scg.make('quadrotor', num_envs=8, output_ring=2, **kw)
    a_env.reset(); b_env.reset()
    act = torch.as_tensor(a_env.action_space.sample(n), device='cuda', dtype=torch.float32)
    kept = []
    for k in range(70):
        o1, r1, d1, _ = a_env.step(act)
        o2, r2, d2, _ = b_env.step(act)
        assert o1.shape == o2.shape == (n, a_env.obs_dim)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(a_env.state, b_env.state)
        kept.append((o2, o2.clone()))
        if k >= 2:
            assert torch.equal(kept[k - 1][0], kept[k - 1][1]) and torch.equal(kept[k - 2][0], kept[k - 2][1])   # K - 1 steps of validity
        if k >= 3:
            assert kept[k - 3][0].data_ptr() == o2.data_ptr()                                                   # ... then reused
    b_env.reset()                                      # an in-place writer first takes a private copy of what was handed out
    assert torch.equal(kept[-1][0], kept[-1][1])
    a_env.reset()
    for k in range(5):                                 # ... and the ring carries on behind it
        o1, r1, d1, _ = a_env.step(act)
        o2, r2, d2, _ = b_env.step(act)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(a_env.state, b_env.state)
    a_env.close(); b_env.close()


def test_checkpoint_of_another_residual_format_is_refused(gpu):
    """int16 ('attitude') and float32 ('attitude_f32') residual planes must not be loaded into each other."""
    kw = _common(dict(quad_type=3, task='traj_tracking', cost='quadratic', info_level='minimal'))
    a = scg.make('quadrotor', num_envs=128, compensated_state='attitude', **kw)
    b = scg.make('quadrotor', num_envs=128, compensated_state='attitude_f32', **kw)
    a.reset(); b.reset()
    act = torch.as_tensor(a.action_space.sample(128), device='cuda', dtype=torch.float32)
    for _ in range(5):
        a.step(act); b.step(act)
    with pytest.raises(ValueError, match='differently configured'):
        a.load_state_dict(b.state_dict())
    with pytest.raises(ValueError, match='differently configured'):
        b.load_state_dict(a.state_dict())
    a2 = scg.make('quadrotor', num_envs=128, compensated_state='attitude', **kw)
    a2.load_state_dict(a.state_dict())
    a.step(act); a2.step(act)
    assert torch.equal(a.state, a2.state) and torch.equal(a.residual_planes, a2.residual_planes)
    for e in (a, b, a2):
        e.close()


def test_calls_on_the_handles_device_pass_the_device_guard(gpu):
    """One handle per device: the Python Engine makes every call with its device current, so the guard never fires there; a raw
    C-ABI call is refused only when another device is current (exercised by the host-shim driver: one GPU here)."""
    from safe_control_gym_amd import _capi
    env = scg.make('quadrotor', quad_type=1, num_envs=64, **_common(dict(info_level='minimal')))
    env.reset()
    act = torch.zeros((64, 1), device='cuda')
    env.step(act)
    env.rollout(torch.zeros((2, 1, env.plane_stride), device='cuda'))
    env.linearize()
    lib, eng = _capi.load(), env._engine
    assert lib.scg_step_wait(eng.handle, None) == 0
    q = eng.queue_stream()
    assert q.cuda_stream != 0 and eng.queue_stream() is q
    env.close()


def test_pipelined_loop_is_capturable_in_a_hip_graph(gpu):
    """The event fork / join of step_async / step_wait and the launches on the shards' queues are capturable: K steps of the
    pipelined loop (policy kernels included) recorded once with torch.cuda.graph and replayed equal K eager steps of the plain env.
    (The step serials are baked into the nodes: a SECOND replay would repeat the first one's reset draws -- scg_rollout's own
    graph replay passes the serial through a device word for that reason.)"""
    kw = _common(dict(quad_type=3, task='traj_tracking', cost='quadratic', info_level='minimal'))
    n, warm, K = 2048, 3, 70                         # 60-step episodes: auto-resets inside the captured stretch
    plain = scg.make('quadrotor', num_envs=n, **kw)
    venv = make_vec_envs(partial(scg.make, 'quadrotor', **kw), batch_size=n, num_pipelines=2)
    envs = venv.envs
    lo = torch.as_tensor(plain.action_space.low, device='cuda').reshape(-1, 1)
    hi = torch.as_tensor(plain.action_space.high, device='cuda').reshape(-1, 1)
    bias, gain = 0.5 * (lo + hi), 0.1 * (hi - lo)
    acts = [torch.zeros((4, e.plane_stride), device='cuda') for e in envs]          # the policy's output planes, one buffer per shard
    act_plain = torch.zeros((4, plain.plane_stride), device='cuda')

    def policy(obs, out, m):                                                        # capture-safe: no allocation, no host sync
        torch.addcmul(bias, obs.T[:4], gain, out=out[:, :m])
        return out
    obs = plain.reset()
    pobs = [e.reset() for e in envs]

    def pipelined(steps, on_queue):
        cur = torch.cuda.current_stream()
        qs = [e.pipeline_stream if on_queue else cur for e in envs]
        for q in qs:
            q.wait_stream(cur)
        for _ in range(steps):
            for p, e in enumerate(envs):
                with torch.cuda.stream(qs[p]):
                    e.step_async(policy(pobs[p], acts[p], e.num_envs))
            for p, e in enumerate(envs):
                with torch.cuda.stream(qs[p]):
                    pobs[p] = e.step_wait()[0]
        for q in qs:
            cur.wait_stream(q)
    for on_queue in (False, True):
        for _ in range(warm):
            obs = plain.step(policy(obs, act_plain, n))[0]
        pipelined(warm, on_queue)                                                   # eager: queues and events exist before the capture
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        import gc
        gc.collect(); gc.disable()                                                  # no finaliser (scg_destroy of an old env) inside the capture
        try:
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    pipelined(K, on_queue)
        finally:
            gc.enable()
        torch.cuda.current_stream().wait_stream(side)
        graph.replay()
        for _ in range(K):
            obs = plain.step(policy(obs, act_plain, n))[0]
        torch.cuda.synchronize()
        _same(torch.cat([e.state for e in envs]), plain.state, 'state after the replay (on_queue=%s)' % on_queue)
        _same(torch.cat([e.ctrl_step_counter for e in envs]), plain.ctrl_step_counter, 'step counters after the replay')
        _same(torch.cat([e.episode_return for e in envs]), plain.episode_return, 'episode returns after the replay')
        assert int(plain.ctrl_step_counter.max()) < K                                # every env was auto-reset inside the captured stretch
        del graph
    venv.close(); plain.close()


@pytest.mark.parametrize('tier', ['lean', 'action_noise', 'goal_horizon_views'])
def test_captured_loop_replays_equal_eager_stepping(gpu, tier):
    """capture_loop(): K iterations of policy -> step recorded in one HIP graph, the step serials as offsets from a device word
    (scg_step_io.serial_word) rewritten before every replay.  Three replays equal 3 K eager steps of a twin env bit for bit -- through
    auto-resets (their Philox draws are counted by the serial) and, in the noise tier, through in-kernel action noise that is drawn
    from the serial every step: a graph with baked serials would repeat the first replay's numbers."""
    from safe_control_gym_amd.envs.graph_loop import capture_loop
    kw = {'lean': dict(quad_type=3, task='traj_tracking', cost='quadratic', info_level='minimal'),
          'action_noise': dict(quad_type=2, task='stabilization', cost='rl_reward', info_level='minimal',
                               disturbances={'action': [{'disturbance_func': 'white_noise', 'std': 0.002}]}),
          'goal_horizon_views': dict(quad_type=2, task='traj_tracking', cost='rl_reward', obs_goal_horizon=1, copy_outputs=False)}[tier]
    kw = _common(kw)
    n, K, warm = 3000, 40, 2
    eager = scg.make('quadrotor', num_envs=n, **kw)
    single = scg.make('quadrotor', num_envs=n, **kw)
    venv = make_vec_envs(partial(scg.make, 'quadrotor', **kw), batch_size=n, num_pipelines=2)
    nu = eager.action_dim
    lo = torch.as_tensor(eager.action_space.low, device='cuda').reshape(-1, 1)
    hi = torch.as_tensor(eager.action_space.high, device='cuda').reshape(-1, 1)
    bias, gain = 0.5 * (lo + hi), 0.1 * (hi - lo)

    def policy_into(env):
        out = torch.zeros((nu, env.plane_stride), device='cuda')

        def policy(obs):                                            # capture-safe: writes a preallocated plane buffer
            torch.addcmul(bias, obs.T[:nu], gain, out=out[:, :env.num_envs])
            return out
        return policy
    pe, ps, pv = policy_into(eager), policy_into(single), [policy_into(e) for e in venv.envs]
    obs_e, obs_s = eager.reset(), single.reset()
    obs_v = [e.reset() for e in venv.envs]

    def body_single(k):
        single.step(ps(obs_s))                                      # views: obs_s is the same tensor in every step

    def body_venv(k):
        for p, e in enumerate(venv.envs):
            with torch.cuda.stream(e.pipeline_stream):
                e.step_async(pv[p](obs_v[p]))
        for p, e in enumerate(venv.envs):
            with torch.cuda.stream(e.pipeline_stream):
                e.step_wait()

    def fork_join(body):                                            # the shards' queues join the recording through the caller's stream
        def wrapped(k):
            cur = torch.cuda.current_stream()
            for e in venv.envs:
                e.pipeline_stream.wait_stream(cur)
            body(k)
            for e in venv.envs:
                cur.wait_stream(e.pipeline_stream)
        return wrapped
    loop_s = capture_loop(single, body_single, K, warmup=warm)
    loop_v = venv.capture_loop(fork_join(body_venv), K, warmup=warm)
    assert loop_s.serials == [K] and loop_v.serials == [K, K]
    for _ in range(warm):
        obs_e = eager.step(pe(obs_e))[0]
    resets = 0
    for r in range(3):
        loop_s.replay()
        loop_v.replay()
        for _ in range(K):
            obs_e, _, done, _ = eager.step(pe(obs_e))
            resets += int(done.sum())
        torch.cuda.synchronize()
        for other, what in ((single.state, 'one env'), (torch.cat([e.state for e in venv.envs]), 'two pipelined shards')):
            _same(other, eager.state, '%s: state after replay %d' % (what, r))
        _same(single.episode_return, eager.episode_return, 'episode returns after replay %d' % r)
        _same(torch.cat([e.ctrl_step_counter for e in venv.envs]), eager.ctrl_step_counter, 'step counters after replay %d' % r)
        _same(single._obs_view(), eager._obs_view(), 'observation after replay %d' % r)
    assert resets >= n
    # eager steps and recorded replays mix freely: the env's serial runs on
    obs_e = eager.step(pe(obs_e))[0]
    single.step(ps(obs_s))
    _same(single.state, eager.state, 'an eager step after the replays')
    loop_s.replay()
    for _ in range(K):
        obs_e = eager.step(pe(obs_e))[0]
    _same(single.state, eager.state, 'a replay after an eager step')
    # what cannot be recorded says so
    fresh = scg.make('quadrotor', quad_type=2, num_envs=64, **_common({}))
    fresh.reset()
    with pytest.raises(ValueError, match='steps in place'):
        capture_loop(fresh, lambda k: None, 4, warmup=0)
    with pytest.raises(RuntimeError, match='inside capture_loop'):
        capture_loop(single, lambda k: single.reset(), 2, warmup=0)
    assert single._capture_base is None and single._engine._io.serial_word is None      # a failed recording leaves no trace
    for e in (eager, single, fresh):
        e.close()
    venv.close()


def test_capture_loop_refusals_leave_the_envs_usable(gpu):
    """A recorded stretch that cannot be replayed is refused when it is recorded, and the env steps eagerly afterwards as if nothing
    had happened: (a) an odd number of updating normaliser calls (the statistics records are double-buffered on the host);
    (b) a shard with fresh per-step outputs next to one that steps in place."""
    from safe_control_gym_amd.envs.graph_loop import capture_loop
    from safe_control_gym_amd.math_and_models.normalization import MeanStdNormalizer
    kw = _common(dict(quad_type=2, task='stabilization', cost='rl_reward', info_level='minimal'))
    env, twin = scg.make('quadrotor', num_envs=512, **kw), scg.make('quadrotor', num_envs=512, **kw)
    env.reset(); twin.reset()
    act = torch.as_tensor(env.action_space.sample(512), device='cuda', dtype=torch.float32)
    norm = MeanStdNormalizer(shape=(env.obs_dim,))
    planes = torch.empty((env.obs_dim, env.plane_stride), device='cuda')

    def body(k):
        obs, _, _, _ = env.step(act)
        norm(obs, out=planes)
    with pytest.raises(ValueError, match='odd number'):
        capture_loop(env, body, 3, warmup=0)
    cur = norm.rms._cur
    loop = capture_loop(env, body, 4, warmup=0)                 # an even stretch is fine, and starts from the same record
    assert norm.rms._cur == cur
    loop.replay()
    for _ in range(4):
        o2 = twin.step(act)[0]
    _same(env.step(act)[0], twin.step(act)[0], 'the step after a refused and a replayed recording')
    fresh = scg.make('quadrotor', num_envs=512, **_common(dict(quad_type=2, task='stabilization', cost='rl_reward')))
    fresh.reset()
    with pytest.raises(ValueError, match='steps in place'):
        capture_loop([env, fresh], lambda k: (env.step(act), fresh.step(act)), 2, warmup=0)
    _same(env.step(act)[0], twin.step(act)[0], 'the step after a refused two-env recording')   # env left recording mode again
    env.close(); twin.close(); fresh.close()


BASELINE_SIZES = {
    'c4_2p20': ('quadrotor', 1 << 20, dict(quad_type=3, task='traj_tracking', cost='quadratic', info_level='minimal', episode_len_sec=5)),
    'c3_2p18': ('quadrotor', 1 << 18, dict(quad_type=2, task='stabilization', cost='rl_reward', constraints=C3_CONSTRAINTS,
                                           info_level='minimal', episode_len_sec=5)),
    'c2_2p16': ('cartpole', 1 << 16, dict(task='stabilization', cost='quadratic', info_level='minimal', ctrl_freq=50, pyb_freq=50,
                                          episode_len_sec=5)),
}


@pytest.mark.parametrize('case', list(BASELINE_SIZES))
def test_round4_forms_at_baseline_sizes(gpu, case):
    """At BASELINE.json's full sizes (C4 2^20 -- the natural two-queue split --, C3 2^18, C2 2^16): the trajectory sink filled by
    per-step launches and by the fused launch, and two pipelined shards, reproduce the plain step() loop bit for bit (a
    size-independent property: equality of every observation / reward / done of every step).  A short stretch with a high
    termination rate (wide random actions) so that auto-resets occur inside it."""
    name, n, kw = BASELINE_SIZES[case]
    kw = _common(kw)
    steps = 12
    plain, sunk, fused = (scg.make(name, num_envs=n, **kw) for _ in range(3))
    venv = make_vec_envs(partial(scg.make, name, **kw), batch_size=n, num_pipelines=2)
    for e in (plain, sunk, fused):
        e.reset()
    pobs = [venv.reset(pipeline=p)[0] for p in range(2)]
    ld, nu = plain.plane_stride, plain.action_dim
    g = torch.Generator(device='cuda'); g.manual_seed(17)
    lo = torch.as_tensor(plain.action_space.low, device='cuda')[None, :, None]
    hi = torch.as_tensor(plain.action_space.high, device='cuda')[None, :, None]
    acts = torch.rand((4, nu, ld), generator=g, device='cuda') * (hi - lo) * 1.6 + lo - 0.3 * (hi - lo)
    ts, tf = sunk.trajectory_buffer(steps), fused.trajectory_buffer(steps)
    sunk.rollout(acts, steps=steps, out=ts)
    fused.rollout(acts, steps=steps, fused=True, out=tf)
    finished = 0
    for k in range(steps):
        obs, rew, done, info = plain.step(acts[k % 4])
        rows = acts[k % 4][:, :n].T.contiguous()
        for p, (a, b) in enumerate(venv.bounds):
            venv.step_async(rows[a:b], pipeline=p)
        outs = [venv.step_wait(pipeline=p) for p in range(2)]
        for q, what in enumerate(('obs', 'reward', 'done')):
            _same(torch.cat([o[q] for o in outs]), (obs, rew, done)[q], 'pipelined shards: %s at step %d' % (what, k))
        for traj, form in ((ts, 'sink'), (tf, 'fused sink')):
            _same(traj.obs[k], obs, '%s: obs at step %d' % (form, k))
            _same(traj.reward[k], rew, '%s: reward at step %d' % (form, k))
            _same(traj.done[k], done, '%s: done at step %d' % (form, k))
            if plain.constraints is not None:
                _same(traj.constraint_values[k], info['constraint_values'], '%s: constraint values at step %d' % (form, k))
        finished += int(done.sum())
    assert finished > 0
    for e in (sunk, fused):
        _same(e.state, plain.state, 'final state')
        _same(e.ctrl_step_counter, plain.ctrl_step_counter, 'final step counters')
    for e in (plain, sunk, fused):
        e.close()
    venv.close()


@pytest.mark.parametrize('extra', [{}, dict(obs_goal_horizon=1),
                                   dict(disturbances={'observation': [{'disturbance_func': 'white_noise', 'std': 0.01}]})])
def test_output_pool_recycles_only_what_nobody_holds(gpu, extra):
    """output_pool (default of the fresh-output mode): step() recycles an output set once the caller holds no reference to any of it
    -- same numbers as allocating per step, and a tensor that anybody can still read (the object itself, a view derived from it,
    ``info['n']``'s ``done``) is never written again; a caller that keeps everything falls back to allocation."""
    n = 1000
    kw = _common(dict(quad_type=3, task='traj_tracking', cost='quadratic', **extra))
    a_env = scg.make('quadrotor', num_envs=n, output_pool=0, **kw)
    b_env = scg.make('quadrotor', num_envs=n, **kw)
    with pytest.raises(ValueError, match='output_pool'):
        scg.make('quadrotor', num_envs=8, output_pool=2, **kw)
    a_env.reset(); b_env.reset()
    assert a_env._pool is None and b_env._pool is not None
    act = torch.as_tensor(a_env.action_space.sample(n), device='cuda', dtype=torch.float32)
    seen = set()
    for k in range(30):                                                # a loop that holds one step's outputs at a time
        o1 = a_env.step(act)[0]
        o2, r2, d2, info = b_env.step(act)
        assert torch.equal(o1, o2)
        seen.add(o2.data_ptr())
    assert len(b_env._pool) <= 3 and len(seen) <= 3                    # ... lives on two or three sets
    del o2, r2, d2, info
    kept, ptrs = [], set()
    prev = None
    for k in range(90):
        o1, r1, d1, _ = a_env.step(act)
        o2, r2, d2, info = b_env.step(act)
        assert o2.shape == (n, a_env.obs_dim) and r2.shape == (n,) and d2.shape == (n,) and d2.dtype == torch.bool
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(a_env.state, b_env.state)
        ptrs.add(o2.data_ptr())
        if prev is not None:
            assert prev[0].data_ptr() != o2.data_ptr() and torch.equal(prev[0], prev[1])      # the usual loop: obs, then next_obs
        prev = (o2, o2.clone())
        if k % 9 == 0:
            kept.append(('obs', o2, o2.clone()))                       # the object itself
        elif k % 9 == 1:
            kept.append(('view', o2[:, :3], o2[:, :3].clone()))        # only a view derived from it (the object is dropped)
        elif k % 9 == 2:
            kept.append(('reward', r2, r2.clone()))                    # only the reward
        elif k % 9 == 3:
            kept.append(('info', info, d2.clone()))                    # only info: info['n'] reads this step's done
        del o2, r2, d2, info
    for kind, held, copy in kept:
        assert torch.equal(held['n']._done if kind == 'info' else held, copy), kind
    assert len(b_env._pool) <= 8
    assert len(ptrs) < 60                                              # sets were recycled (40 of the 90 steps pinned theirs)
    # a caller that keeps everything: nothing is overwritten, step() allocates again
    hoard = []
    for k in range(40):
        o1 = a_env.step(act)[0]
        o2, r2, d2, _ = b_env.step(act)
        assert torch.equal(o1, o2)
        hoard.append((o2, r2, d2, o2.clone(), r2.clone(), d2.clone()))
    for o, r, d, oc, rc, dc in hoard:
        assert torch.equal(o, oc) and torch.equal(r, rc) and torch.equal(d, dc)
    assert len({h[0].data_ptr() for h in hoard}) == 40
    del hoard, kept, prev
    a_env.reset(); b_env.reset()                                       # in-place writers take private copies first
    for k in range(40):
        o1, r1, d1, _ = a_env.step(act)
        o2, r2, d2, _ = b_env.step(act)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    a_env.close(); b_env.close()


def test_output_pool_with_asynchronous_steps(gpu):
    """step_async / step_wait in the fresh-output mode: what the pending kernel reads and writes stays referenced until the join."""
    n = 512
    kw = _common(dict(quad_type=3, task='traj_tracking', cost='quadratic'))
    a_env = scg.make('quadrotor', num_envs=n, output_pool=0, **kw)
    b_env = scg.make('quadrotor', num_envs=n, **kw)
    a_env.reset(); b_env.reset()
    act = torch.as_tensor(a_env.action_space.sample(n), device='cuda', dtype=torch.float32)
    last = None
    for k in range(50):
        o1, r1, d1, _ = a_env.step(act)
        b_env.step_async(act)
        o2, r2, d2, _ = b_env.step_wait()
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
        if last is not None:
            assert torch.equal(last[0], last[1])
        last = (o2, o2.clone())
    a_env.close(); b_env.close()
