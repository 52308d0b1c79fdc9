"""CPU-only tests of the host-side mirror of the reference interface (no GPU compute calls)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import safe_control_gym_amd as scg
from safe_control_gym_amd import _capi
from safe_control_gym_amd.envs.benchmark_env import Cost, Task, Environment
from safe_control_gym_amd.envs.constraints import (BoundedConstraint, DefaultConstraint, QuadraticContstraint,
                                                   ConstraintList, GENERAL_CONSTRAINTS)
from safe_control_gym_amd.envs.spaces import Box
from oracle.ref_env import RefEnv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_surface():
    """register / make / spec / get_config as in utils/registration.py:138-161."""
    cfg = scg.get_config('quadrotor')
    assert cfg['ctrl_freq'] == 60 and cfg['pyb_freq'] == 240 and cfg['quad_type'] == 2
    assert scg.spec('quadrotor').id == 'quadrotor'
    with pytest.raises(Exception, match='Cannot re-register'):
        scg.register(id='quadrotor', entry_point='x:y')
    with pytest.raises(Exception, match='Key not found'):
        scg.spec('nope')
    env = scg.make('quadrotor', num_envs=3)
    assert env.spec.id == 'quadrotor' and env.NAME == 'quadrotor' and env.num_envs == 3
    assert scg.make('cartpole', num_envs=2).NAME == 'cartpole'


def test_enums_match_reference_values():
    assert Cost('rl_reward') == Cost.RL_REWARD and Cost('quadratic') == Cost.QUADRATIC
    assert Task('stabilization') == Task.STABILIZATION and Task('traj_tracking') == Task.TRAJ_TRACKING
    assert Environment.CARTPOLE == 'cartpole' and Environment.QUADROTOR == 'quadrotor'


@pytest.mark.parametrize('quad_type,nx,nu', [(1, 2, 1), (2, 6, 2), (3, 12, 4)])
def test_quadrotor_spaces_and_goals_match_oracle(quad_type, nx, nu):
    ti = {'stabilization_goal': [0.3, 1.1, 0.7], 'stabilization_goal_tolerance': 0.05}
    for task in ('stabilization', 'traj_tracking'):
        kw = dict(task=task, ctrl_freq=60, pyb_freq=240, cost='quadratic')
        if task == 'stabilization':
            kw['task_info'] = ti
        env = scg.make('quadrotor', quad_type=quad_type, num_envs=2, **kw)
        ref = RefEnv(system=quad_type, **kw)
        assert env.state_dim == nx and env.action_dim == nu and env.obs_dim == nx
        assert np.array_equal(env.action_space.low.astype(np.float64), ref.action_space_low)
        assert np.array_equal(env.action_space.high.astype(np.float64), ref.action_space_high)
        assert np.array_equal(env.state_space.low.astype(np.float64), ref.state_space_low)
        assert np.array_equal(env.state_space.high.astype(np.float64), ref.state_space_high)
        assert env.action_space.low.dtype == np.float32
        assert np.max(np.abs(env.X_GOAL - ref.X_GOAL)) < 1e-15
        assert np.array_equal(env.U_GOAL, ref.U_GOAL) and np.array_equal(env.U_EQ, env.U_GOAL)
        assert env.CTRL_STEPS == 300 and env.PYB_STEPS_PER_CTRL == 4 and env.CTRL_TIMESTEP == 1 / 60


@pytest.mark.parametrize('traj', ['circle', 'figure8', 'square'])
@pytest.mark.parametrize('plane', ['zx', 'xy', 'xz'])
def test_reference_tables_match_oracle(traj, plane):
    ti = dict(scg.make('quadrotor', num_envs=1).TASK_INFO)
    ti.update(trajectory_type=traj, trajectory_plane=plane, num_cycles=2, trajectory_position_offset=[0.2, -0.1],
              trajectory_scale=0.7)
    for qt in (1, 2, 3):
        kw = dict(task='traj_tracking', task_info=ti, ctrl_freq=50, pyb_freq=50, episode_len_sec=7)
        env = scg.make('quadrotor', quad_type=qt, num_envs=1, **kw)
        ref = RefEnv(system=qt, **kw)
        assert env.X_GOAL.shape == ref.X_GOAL.shape == (350, env.state_dim)
        assert np.max(np.abs(env.X_GOAL - ref.X_GOAL)) < 1e-14


def test_obs_goal_horizon_widens_observation_space():
    env = scg.make('quadrotor', quad_type=2, num_envs=1, task='traj_tracking', cost='rl_reward', obs_goal_horizon=3)
    assert env.obs_dim == 6 * 4 and env.state_dim == 6
    env = scg.make('quadrotor', quad_type=2, num_envs=1, task='stabilization', cost='rl_reward', obs_goal_horizon=3)
    assert env.obs_dim == 12
    env = scg.make('quadrotor', quad_type=2, num_envs=1, task='traj_tracking', cost='quadratic', obs_goal_horizon=3)
    assert env.obs_dim == 6                         # quadrotor.py:794: only for rl_reward


def test_error_conventions():
    """Exception types of the reference ('Error conventions')."""
    with pytest.raises(ValueError, match='not divisible'):
        scg.make('quadrotor', num_envs=1, pyb_freq=100, ctrl_freq=60)
    with pytest.raises(ValueError, match='init_state incorrect format'):
        scg.make('quadrotor', num_envs=1, init_state=[0, 0])
    with pytest.raises(ValueError, match='inertial_prop incorrect format'):
        scg.make('quadrotor', num_envs=1, inertial_prop=[1, 2, 3])
    with pytest.raises(ValueError, match='info_mse_metric_state_weight'):
        scg.make('quadrotor', num_envs=1, info_mse_metric_state_weight=[1, 0])
    env = scg.make('quadrotor', num_envs=4)
    with pytest.raises(RuntimeError, match='reset'):                    # benchmark_env.py:268-275
        env.step(np.zeros((4, 2), dtype=np.float32))
    with pytest.raises(RuntimeError, match='adversary_disturbance does not exist'):
        env.set_adversary_control(np.zeros((4, 2)))
    with pytest.raises(AssertionError):
        scg.make('quadrotor', num_envs=1, constraints=[{'constrained_variable': 'state'}])
    with pytest.raises(AssertionError):
        scg.make('quadrotor', num_envs=1, constraints=[{'constraint_form': 'nope', 'constrained_variable': 'state'}])
    with pytest.raises(AssertionError):
        scg.make('quadrotor', num_envs=1, disturbances={'action': [{'disturbance_func': 'nope'}]})
    with pytest.raises(NotImplementedError):
        scg.make('quadrotor', num_envs=1, cost='competition')


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_engine_fails_loudly_without_gpu():
    env = scg.make('quadrotor', num_envs=4)
    with pytest.raises(_capi.ScgError, match='no ROCm device|ROCm device only'):
        env.reset()


def test_box_space():
    b = Box(low=np.array([-1.0, 0.0]), high=np.array([1.0, 2.0]))
    b.seed(3)
    s = b.sample()
    assert s.dtype == np.float32 and b.contains(s) and b.sample(5).shape == (5, 2)
    assert not b.contains(np.array([2.0, 0.0], dtype=np.float32))


CONSTRAINT_SPECS = [
    [{'constraint_form': 'default_constraint', 'constrained_variable': 'input'},
     {'constraint_form': 'default_constraint', 'constrained_variable': 'state'}],
    [{'constraint_form': 'default_constraint', 'constrained_variable': 'input', 'strict': True},
     {'constraint_form': 'bounded_constraint', 'constrained_variable': 'state', 'active_dims': [0, 2, 4],
      'upper_bounds': [3, 3, 2], 'lower_bounds': [-3, -3, -0.1], 'tolerance': [0.1] * 6}],
    [{'constraint_form': 'linear_constraint', 'constrained_variable': 'state', 'A': [[1.0, 0.5, 0, 0, 0, 0], [0, 0, 1.0, -1.0, 0.2, 0]],
      'b': [1.0, 2.0]},
     {'constraint_form': 'quadratic_constraint', 'constrained_variable': 'input', 'P': [[2.0, 0.1], [0.1, 1.0]], 'b': 0.5},
     {'constraint_form': 'linear_constraint', 'constrained_variable': 'input_and_state',
      'A': [[1, 0, 0, 0, 0, 0, 1.0, -1.0]], 'b': [0.3]}],
]


@pytest.mark.parametrize('specs', CONSTRAINT_SPECS)
def test_constraint_rows_reproduce_oracle_values(specs):
    """The rows handed to scg_set_constraints, evaluated in float64 on the host, equal the oracle's values."""
    env = scg.make('quadrotor', quad_type=2, num_envs=1, constraints=specs)
    ref = RefEnv(system=2, constraints=specs)
    cl = env.constraints
    assert cl.num_constraints == ref.constraints.num_constraints
    rows, n_rows, pmats, n_p = cl.compile_rows()
    rng = np.random.default_rng(0)
    P = np.array(list(pmats)).reshape(-1, 16, 16) if n_p else None
    for _ in range(20):
        x, a = rng.normal(size=6), rng.uniform(0.0, 0.35, size=2)
        ref.state, ref.current_raw_input_action = x, a
        expect = ref.constraints.get_values(ref)
        v = np.zeros(16)
        v[:6], v[6:8] = x, a
        got = []
        for r in range(n_rows):
            row = rows[r]
            if row.kind == 0:
                got.append(np.dot(np.array(list(row.coef)), v) - row.b)
            else:
                got.append(v @ P[row.p_index] @ v - row.b)
        assert np.max(np.abs(np.array(got) - expect)) < 1e-7       # oracle rounds to 8 decimals
    for k, con in enumerate(cl.constraints):
        assert con.strict == ref.constraints.constraints[k].strict


GOAL3 = {'stabilization_goal': [0, 0, 1], 'stabilization_goal_tolerance': 0.05}


def test_constraint_classes_keep_reference_attributes():
    with pytest.raises(IndexError):
        scg.make('quadrotor', quad_type=3, num_envs=1)         # quadrotor.py:277 indexes goal[2] of the 2-D default
    env = scg.make('quadrotor', quad_type=3, num_envs=1, task_info=GOAL3)
    c = BoundedConstraint(env, lower_bounds=[-3, -3, -0.1], upper_bounds=[3, 3, 2], constrained_variable='state',
                          active_dims=[0, 2, 4])
    assert c.num_constraints == 6 and c.dim == 3 and c.constraint_filter.shape == (3, 12)
    assert np.array_equal(c.A, np.vstack([-np.eye(3), np.eye(3)])) and np.array_equal(c.b, [3, 3, 0.1, 3, 3, 2])
    d = DefaultConstraint(env, constrained_variable='input')
    assert d.num_constraints == 8 and np.allclose(d.lower_bounds, env.action_space.low)
    assert d.sym_func(env.U_GOAL).shape == (8,)
    q = QuadraticContstraint(env, P=np.eye(4), b=1.0, constrained_variable='input')
    assert q.num_constraints == 1
    cl = ConstraintList([c, d, q])
    assert cl.num_constraints == 15 and list(cl.constraint_indices) == [6, 14] and len(cl) == 3
    assert cl.num_state_constraints == 6 and cl.num_input_constraints == 9
    assert set(GENERAL_CONSTRAINTS) == {'linear_constraint', 'quadratic_constraint', 'bounded_constraint', 'default_constraint'}
    with pytest.raises(ValueError, match='tolerance dimension'):
        BoundedConstraint(env, lower_bounds=[0], upper_bounds=[1], constrained_variable='state', active_dims=[4],
                          tolerance=[0.1])


def test_symbolic_model_surface():
    env = scg.make('quadrotor', quad_type=3, num_envs=1, task_info=GOAL3)
    m = env.symbolic
    assert (m.nx, m.nu, m.ny) == (12, 4, 12) and m.dt == env.CTRL_TIMESTEP
    x = np.zeros(12); x[4] = 1.0
    assert np.allclose(m.fc_func(x=x, u=env.U_GOAL)['f'], 0.0, atol=1e-14)
    assert np.allclose(m.fd_func(x0=x, p=env.U_GOAL)['xf'], x, atol=1e-14)
    J = m.df_func(x=x, u=env.U_GOAL)
    assert J['dfdx'].shape == (12, 12) and J['dfdu'].shape == (12, 4)
    assert J['dfdx'][1, 7] == pytest.approx(9.8)                    # x_ddot = g * theta near hover
    out = m.loss(x=x, u=env.U_GOAL, Xr=np.zeros(12), Ur=env.U_GOAL, Q=np.eye(12), R=np.eye(4))
    assert out['l'] == pytest.approx(0.5) and out['l_x'][4] == pytest.approx(1.0)


def test_c_abi_header_matches_library_and_binding():
    """Every function declared in include/scg_rollout.h is exported by the .so and bound by _capi."""
    header = open(os.path.join(ROOT, 'include', 'scg_rollout.h')).read()
    declared = set(re.findall(r'^(?:int|const char\*)\s+(scg_\w+)\s*\(', header, flags=re.M))
    assert {'scg_create', 'scg_destroy', 'scg_set_reference', 'scg_set_constraints', 'scg_reset', 'scg_step',
            'scg_rollout', 'scg_eval_constraints', 'scg_last_error', 'scg_version'} <= declared
    lib = _capi.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_capi.EXPORTS)
    nm = subprocess.run(['nm', '-D', '--defined-only', _capi.library_path()], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (scg_\w+)', nm))
    assert declared <= exported
    assert lib.scg_version() == _capi.SCG_ABI_VERSION
    assert lib.scg_sizeof_config() == ctypes.sizeof(_capi.ScgConfig)
    assert lib.scg_sizeof_step_io() == ctypes.sizeof(_capi.ScgStepIO)
    assert lib.scg_sizeof_constraint_row() == ctypes.sizeof(_capi.ScgConstraintRow)
    assert lib.scg_sizeof_disturbance() == ctypes.sizeof(_capi.ScgDisturbance)
    for macro, val in (('SCG_MAX_NX', _capi.SCG_MAX_NX), ('SCG_MAX_NU', _capi.SCG_MAX_NU), ('SCG_MAX_NV', _capi.SCG_MAX_NV),
                       ('SCG_MAX_CONSTRAINT_ROWS', _capi.SCG_MAX_CONSTRAINT_ROWS), ('SCG_ABI_VERSION', _capi.SCG_ABI_VERSION),
                       ('SCG_MAX_DISTURBANCES', _capi.SCG_MAX_DISTURBANCES)):
        assert int(re.search(r'#define %s (\d+)' % macro, header).group(1)) == val


def test_library_override_is_explicit_and_fails_loudly(monkeypatch, tmp_path):
    """SCG_ROLLOUT_LIB (read by the Python binding, never by the library) points the binding at another build of the SAME ABI --
    the A/B switch of tools/tune.py.  It is the only such switch: a missing file raises (no fallback to the in-tree library), a
    library of another ABI version is refused by load()."""
    default = _capi.library_path()
    assert default == os.path.join(ROOT, 'safe_control_gym_amd', 'lib', 'libscg_rollout.so')
    monkeypatch.setenv('SCG_ROLLOUT_LIB', str(tmp_path / 'absent.so'))
    assert _capi.library_path() == str(tmp_path / 'absent.so')
    monkeypatch.setattr(_capi, '_LIB', None)
    with pytest.raises(_capi.ScgError, match='missing'):
        _capi.load()
    # the same file under another name loads and is the library that answers
    import shutil
    shutil.copy(default, tmp_path / 'copy.so')
    monkeypatch.setenv('SCG_ROLLOUT_LIB', str(tmp_path / 'copy.so'))
    monkeypatch.setattr(_capi, '_LIB', None)
    lib = _capi.load()
    assert lib._name == str(tmp_path / 'copy.so') and lib.scg_version() == _capi.SCG_ABI_VERSION
    monkeypatch.setattr(_capi, '_LIB', None)


def test_c_abi_rejects_bad_arguments_without_gpu_work():
    lib = _capi.load()
    h = ctypes.c_void_p()
    cfg = _capi.ScgConfig()
    assert lib.scg_create(None, ctypes.byref(h)) == -1
    cfg.struct_size = 7
    assert lib.scg_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b'struct_size' in lib.scg_last_error()
    assert lib.scg_step(None, None, 0, None) == -1
    assert lib.scg_destroy(None) == 0
    # the next-row entry points validate before they launch anything, too
    assert lib.scg_set_disturbances(None, None, 0) == -1
    assert lib.scg_linearize(None, None, None, None, None, None, 0.0, 0, 0, None) == -1
    assert lib.scg_model_eval(None, None, None, None, None, None, 0, 0, None) == -1
    assert lib.scg_quadratic_loss(None, None, None, None, None, None, 0, None, 0, None, None, None, 0, 0, None) == -1
    assert lib.scg_mean_std_normalize(None, None, 3, 8, 8, 4, None, None, None, 10.0, 1e-8, None) == -1
    assert lib.scg_reward_normalize(None, None, None, None, 4, None, None, None, 0.99, 10.0, 1e-8, 0, None) == -1
    assert lib.scg_gae(None, None, None, None, None, None, None, None, None, None, 4, 4, 4, 0.99, 0.95, 1, None) == -1
    assert b'null' in lib.scg_last_error()


def test_shard_bounds_partition():
    from safe_control_gym_amd.parallel import shard_bounds
    for total, world in ((8388608, 8), (10, 4), (7, 8), (1, 1)):
        spans = [shard_bounds(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[r][1] == spans[r + 1][0] for r in range(world - 1))
        assert [hi - lo for lo, hi in spans] == [len(c) for c in np.array_split(np.arange(total), world)]


def test_bench_cli_contract_and_loud_failure_without_gpu():
    """bench.py keeps the driver's flags (--gpus / --steps / --warmup) and, on a box without a ROCm device, fails loudly
    instead of falling back to anything (the product has no CPU path)."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, 'bench.py')
    out = subprocess.run([sys.executable, bench, '--help'], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and all(f in out.stdout for f in ('--gpus', '--steps', '--warmup'))
    import torch
    if not torch.cuda.is_available():
        run = subprocess.run([sys.executable, bench, '--steps', '2', '--warmup', '1', '--envs-per-gpu', '64', '--no-cpu-baseline'],
                             capture_output=True, text=True, timeout=300)
        assert run.returncode != 0 and not run.stdout.strip().startswith('{')


def test_sanitizer_builds_of_the_cpu_side_are_clean():
    """SURVEY.md section 5 (build: -fsanitize=address host shim): GPU sanitizers are not available on this pool, so the
    sanitizers run on the CPU builds -- (a) oracle/c, every system and the edge batch sizes, on exact-size heap buffers;
    (b) the C-ABI validation layer (scg_capi.hip compiled as host C++ against tests/hostshim's stand-in HIP runtime):
    every argument check, the reference / constraint / disturbance table compilers at their limits, and scg_rollout's error
    paths (a launch failing at any point between fork and join still joins the second queue).  Both under
    -fsanitize=address,undefined -fno-sanitize-recover=all: any report aborts with a non-zero exit code."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None or shutil.which('g++') is None:
        pytest.skip('no host compiler on this box (the GPU box runs the prebuilt product only)')
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle', 'c'), '-s', 'asan'], stderr=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, 'oracle', 'c', '_asan', 'scg_oracle_asan')], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and 'asan driver: ok' in out.stdout, out.stdout + out.stderr[-3000:]
    subprocess.check_call(['bash', os.path.join(ROOT, 'tests', 'hostshim', 'build.sh')], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, 'tests', 'hostshim', '_build', 'capi_sanitize')], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and 'capi sanitize driver: ok' in out.stdout, out.stdout + out.stderr[-3000:]


def test_reference_name_aliases_and_registry_surface():
    """Names a caller of the reference imports that have no work of their own here: the registry object and `load`
    (registration.py:12-22,89-135), SymbolicModel, the single-env RecordEpisodeStatistics, random_sample, render stubs."""
    import torch
    from safe_control_gym_amd.utils import registration as reg
    from safe_control_gym_amd.math_and_models.symbolic_systems import SymbolicModel, AnalyticModel
    from safe_control_gym_amd.envs.env_wrappers.record_episode_statistics import RecordEpisodeStatistics, VecRecordEpisodeStatistics
    from safe_control_gym_amd.envs.env_wrappers.vectorized_env.vec_env import VecEnv
    from safe_control_gym_amd.controllers.ppo.ppo_utils import random_sample
    assert reg.load('safe_control_gym_amd.envs.quadrotor:Quadrotor').NAME == 'quadrotor'
    assert reg.registry.spec('quadrotor') is reg.spec('quadrotor') and 'cartpole' in reg.registry.specs
    assert isinstance(reg.registry.get_config('quadrotor'), dict)
    with pytest.raises(Exception, match='Key not found'):
        reg.make('no_such_env')
    assert SymbolicModel is AnalyticModel and RecordEpisodeStatistics is VecRecordEpisodeStatistics
    env = scg.make('quadrotor', quad_type=2, num_envs=3)
    assert isinstance(env.symbolic, SymbolicModel)
    for name in ('render', 'get_images', 'get_viewer'):
        with pytest.raises(NotImplementedError):
            getattr(VecEnv, name)(object.__new__(VecEnv))
    torch.manual_seed(0)
    for drop_last, sizes in ((True, [4, 4, 4]), (False, [4, 4, 4, 3])):
        batches = list(random_sample(np.arange(15), 4, drop_last))
        assert [len(b) for b in batches] == sizes
        flat = torch.cat(batches).numpy()
        assert len(np.unique(flat)) == len(flat) and set(flat) <= set(range(15))
    con = scg.make('quadrotor', quad_type=2, num_envs=3, constraints=[{'constraint_form': 'default_constraint', 'constrained_variable': 'input'}]).constraints.constraints[0]
    assert con.get_env_constraint_var(env) is None                      # the raw action before the first step, as env.state is None


def test_extend_obs_helper_matches_reference_rule():
    """benchmark_env.py:406-420 batched: rows min(next_step + i, T - 1) for tracking, the goal for stabilisation, obs unchanged
    otherwise (the oracle's extend_obs is pinned to the executed reference method in test_reference_pins.py)."""
    import torch
    from oracle.ref_env import RefEnv
    kw = dict(task='traj_tracking', cost='rl_reward', obs_goal_horizon=3, ctrl_freq=50, pyb_freq=50)
    env = scg.make('quadrotor', quad_type=2, num_envs=4, device='cpu', **kw)
    ref = RefEnv(system=2, **kw)
    rng = np.random.default_rng(0)
    obs = rng.normal(size=(4, 6))
    steps = np.array([0, 7, 248, 10 ** 6])
    got = env.extend_obs(torch.as_tensor(obs, dtype=torch.float32), torch.as_tensor(steps)).numpy()
    for i in range(4):
        ref.state, ref.ctrl_step_counter = obs[i].copy(), int(steps[i]) - 1
        assert np.abs(got[i] - ref._get_observation()).max() <= 1e-6
    kw = dict(task='stabilization', cost='rl_reward', obs_goal_horizon=1)
    env, ref = scg.make('quadrotor', quad_type=2, num_envs=4, device='cpu', **kw), RefEnv(system=2, **kw)
    ref.state, ref.ctrl_step_counter = obs[0].copy(), 3
    assert np.abs(env.extend_obs(torch.as_tensor(obs, dtype=torch.float32), 4).numpy()[0] - ref._get_observation()).max() <= 1e-6
    plain = scg.make('quadrotor', quad_type=2, num_envs=4, device='cpu')
    x = torch.zeros(4, 6)
    assert plain.extend_obs(x, 1) is x


def test_make_vec_envs_cuts_the_batch_into_pipelines():
    """make_vec_envs(num_pipelines=P): contiguous shards (np.array_split sizes as subproc_vec_env.py:28, on whole workgroups when the
    batch allows), each built with its first GLOBAL env index as env_index_base -- the Philox streams of the unsplit batch."""
    from safe_control_gym_amd.envs.env_wrappers.vectorized_env import make_vec_envs, pipeline_bounds, PipelinedDeviceVecEnv, DeviceVecEnv

    assert pipeline_bounds(1 << 20, 2) == [(0, 1 << 19), (1 << 19, 1 << 20)]
    assert pipeline_bounds(3000, 2) == [(0, 1280), (1280, 3000)]
    assert pipeline_bounds(1000, 2) == [(0, 500), (500, 1000)]            # too small for whole workgroups: plain array_split
    assert pipeline_bounds(7, 3) == [(0, 2), (2, 4), (4, 7)]
    assert pipeline_bounds(3, 8) == [(0, 1), (1, 2), (2, 3)]              # never an empty shard
    for n, p in ((1 << 20, 3), (5000, 4), (1537, 2)):
        b = pipeline_bounds(n, p)
        assert b[0][0] == 0 and b[-1][1] == n and all(x[1] == y[0] for x, y in zip(b, b[1:])) and all(hi > lo for lo, hi in b)

    built = []

    class StubEnv:
        observation_space, action_space = 'obs_space', 'act_space'

        def __init__(self, num_envs, env_index_base=0, **kw):
            self.num_envs, self.env_index_base, self.kw = num_envs, env_index_base, kw
            self.calls = []
            built.append(self)

        def step_async(self, a):
            self.calls.append(('async', len(a)))

        def close(self):
            self.calls.append('close')

    one = make_vec_envs(StubEnv, batch_size=64, seed=5)
    assert isinstance(one, DeviceVecEnv) and built[-1].num_envs == 64 and built[-1].kw == {'seed': 5}
    built.clear()
    venv = make_vec_envs(StubEnv, env_configs=[{'ctrl_freq': 60, 'env_index_base': 4096}] * 2, batch_size=3000, seed=5, num_pipelines=2)
    assert isinstance(venv, PipelinedDeviceVecEnv) and venv.num_pipelines == 2 and venv.num_envs == 3000
    assert [(e.num_envs, e.env_index_base) for e in built] == [(1280, 4096), (1720, 4096 + 1280)]
    assert all(e.kw == {'ctrl_freq': 60, 'seed': 5} for e in built)
    assert venv.bounds == [(0, 1280), (1280, 3000)] and venv.observation_space == 'obs_space'
    venv.step_async(list(range(3000)))                                     # whole batch: split by rows
    venv.step_async([0] * 7, pipeline=1)
    assert built[0].calls == [('async', 1280)] and built[1].calls == [('async', 1720), ('async', 7)]
    venv.close()
    assert built[0].calls[-1] == 'close' and built[1].calls[-1] == 'close'
    with pytest.raises(ValueError, match='env_configs differ'):
        make_vec_envs(StubEnv, env_configs=[{'a': 1}, {'a': 2}], batch_size=8, num_pipelines=2)


def test_no_kernel_of_the_library_spills_registers():
    """Build-quality gate on the gfx950 code objects of the in-tree build (csrc/build.sh leaves one object per translation unit in
    lib/obj): the metadata note of every kernel says vgpr_spill_count = 0 and no private (scratch) segment (SGPRs parked in VGPR lanes are
    register moves, not memory, and are allowed).  A kernel
    that spills still passes every numerics test and silently loses a third of its rate (DESIGN.md / TUNING.md: every forced occupancy
    that spilled was slower)."""
    import glob
    import re
    import subprocess
    import tempfile
    llvm = '/opt/rocm/lib/llvm/bin'
    objs = sorted(glob.glob(os.path.join(ROOT, 'safe_control_gym_amd', 'lib', 'obj', '*.o')))
    if not objs or not all(os.path.exists(os.path.join(llvm, t)) for t in ('clang-offload-bundler', 'llvm-objcopy', 'llvm-readelf')):
        pytest.skip('needs the object files of an in-tree build and the ROCm LLVM tools')
    kernels = 0
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            fat = os.path.join(tmp, os.path.basename(obj) + '.fatbin')             # the device code travels in the .hip_fatbin section
            subprocess.run([llvm + '/llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', obj, fat], check=True, capture_output=True)
            if not os.path.exists(fat) or os.path.getsize(fat) == 0:
                continue                                     # host-only unit
            listed = subprocess.run([llvm + '/clang-offload-bundler', '--list', '--type=o', '--input=' + fat], capture_output=True, text=True).stdout
            targets = [t for t in listed.split() if 'gfx950' in t]
            if not targets:
                continue
            co = os.path.join(tmp, os.path.basename(obj) + '.co')
            subprocess.run([llvm + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + fat, '--targets=' + targets[0], '--output=' + co],
                           check=True, capture_output=True)
            notes = subprocess.run([llvm + '/llvm-readelf', '--notes', co], capture_output=True, text=True, check=True).stdout
            for block in notes.split('.agpr_count')[1:]:
                name = re.search(r'\.name:\s+(\S+)', block)
                fields = {k: int(v) for k, v in re.findall(r'\.(vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)', block)}
                if name is None or len(fields) < 3:
                    continue
                kernels += 1
                assert fields['vgpr_spill_count'] == 0 and fields['private_segment_fixed_size'] == 0, (os.path.basename(obj), name.group(1), fields)
    assert kernels > 100                                     # the step kernel's instantiations alone are ~170
