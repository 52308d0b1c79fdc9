"""Batched, device-resident base environment -- host-side mirror of ``BenchmarkEnv``.

Keeps the reference's constructor keywords, attribute names, exception types and
``reset()`` / ``step()`` protocol (/root/reference/safe_control_gym/envs/benchmark_env.py:47-463) but
represents ``num_envs`` independent environments at once: the state of all of them lives in
struct-of-arrays PyTorch-ROCm tensors and one fused HIP kernel launch advances the whole batch
(``csrc/scg_kernels.hip`` through the C ABI in ``include/scg_rollout.h``).

What differs from the reference, by necessity of batching:
  * ``step(action)`` takes ``[N, nu]`` and returns ``obs [N, obs_dim]``, ``rew [N]``, ``done [N]``
    and an ``info`` dict whose values are ``[N, ...]`` tensors under the reference's key names
    (``mse``, ``goal_reached``, ``constraint_values``, ``constraint_violation``,
    ``TimeLimit.truncated``; plus the vector-env keys ``terminal_observation`` and ``episode``).
    ``info['n'][i]`` materialises the reference's per-env dict lazily.
  * environments auto-reset when done, as the reference's vector envs do
    (env_wrappers/vectorized_env/dummy_vec_env.py:41-47); pass ``auto_reset=False`` for the plain
    single-env behaviour.
  * the physics is the reference's own prior model integrated with its own RK4,
    not PyBullet; reset randomisation draws from a counter-based Philox stream, not MT19937.
"""
import functools
import os
from enum import Enum

import numpy as np
import torch

from safe_control_gym_amd import _capi
from safe_control_gym_amd.envs.constraints import create_constraint_list
from safe_control_gym_amd.envs.engine import Engine
from safe_control_gym_amd.utils.recycle import RecyclingPool


class Cost(str, Enum):
    """benchmark_env.py:19-26."""
    RL_REWARD = 'rl_reward'
    QUADRATIC = 'quadratic'
    COMPETITION = 'competition'     # needs contact / ray queries: not on the analytic hot path


class Task(str, Enum):
    """benchmark_env.py:29-35."""
    STABILIZATION = 'stabilization'
    TRAJ_TRACKING = 'traj_tracking'


class Environment(str, Enum):
    """benchmark_env.py:38-44."""
    CARTPOLE = 'cartpole'
    QUADROTOR = 'quadrotor'


_INTEGRATORS = {'rk4': _capi.SCG_RK4, 'euler': _capi.SCG_EULER}
# the reference's Physics enum values (base_aviary.py:34-43): all integrate the analytic model; the *_gnd / *_drag
# variants add the ground-effect / drag terms (quadrotor.py _fill_system_config), pyb_dw has no single-drone effect
_PHYSICS_ALIASES = {'pyb': 'rk4', 'dyn': 'euler', 'pyb_gnd': 'rk4', 'pyb_drag': 'rk4', 'pyb_dw': 'rk4',
                    'pyb_gnd_drag_dw': 'rk4', 'rk4': 'rk4', 'euler': 'euler'}


class LazyInfoList:
    """``info['n']``: the reference's tuple of per-env dicts (dummy_vec_env.py:49), built on demand."""

    def __init__(self, info, done):
        self._info, self._done = info, done

    def __len__(self):
        return int(self._done.shape[0])

    def __getitem__(self, i):
        info, out = self._info, {}
        for k in ('mse', 'goal_reached', 'constraint_values', 'constraint_violation'):
            if k in info:
                v = info[k][i]
                out[k] = v.cpu().numpy() if v.ndim else v.item()
        done = bool(self._done[i])
        if 'TimeLimit.truncated' in info and bool(info['TimeLimit.truncated'][i]):
            out['TimeLimit.truncated'] = True       # benchmark_env.py:460-462 (key absent before the limit)
        if done:
            # DummyVecEnv.step_wait stashes the finished step's info and observation (dummy_vec_env.py:41-47); callers
            # read e.g. info['n'][i]['terminal_info']['TimeLimit.truncated'] (controllers/ppo/ppo.py:265-269)
            out['terminal_info'] = dict(out)
            if 'terminal_observation' in info:
                out['terminal_observation'] = info['terminal_observation'][i].cpu().numpy()
            if 'episode' in info:
                out['episode'] = {'r': float(info['episode']['r'][i]), 'l': float(info['episode']['l'][i])}
        return out

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class RolloutTrajectory:
    """Sink of a multi-step rollout (``scg_trajectory_sink``): time-major SoA planes on the device and their ``[T, N, ...]`` views.

    ``obs [T, N, obs_dim]`` is what ``step()`` returned at each step (for an env that finished: its reset observation, as the
    reference's vector envs return it, dummy_vec_env.py:41-47), ``reward [T, N]``, ``done [T, N]`` (bool); with constraints
    ``constraint_values [T, N, nc]`` (device rows; symmetric constraints folded by ``ConstraintList.fold``) and
    ``constraint_violation [T, N]``; with ``info`` ``mse [T, N]`` and ``truncated [T, N]``.  Rows ``>= filled`` are stale."""

    def __init__(self, env, steps, constraints, info):
        n, ld, od = env.num_envs, env._ld, env.obs_dim
        dev, f32 = env.device, torch.float32
        self.env_signature = (n, ld, od)
        self.steps, self.filled = steps, 0
        self._n = n
        self._planes = {'obs': torch.zeros((steps, od, ld), dtype=f32, device=dev),
                        'reward': torch.zeros((steps, ld), dtype=f32, device=dev),
                        'done': torch.zeros((steps, ld), dtype=torch.bool, device=dev)}
        if constraints:
            if env.constraints is None:
                raise ValueError('[ERROR] the env has no constraints to record.')
            self._planes['constraint_values'] = torch.zeros((steps, env.constraints.num_device_rows, ld), dtype=f32, device=dev)
            self._planes['constraint_violation'] = torch.zeros((steps, ld), dtype=torch.bool, device=dev)
            self._fold = env.constraints.fold if env.constraints.num_device_rows != env.constraints.num_constraints else None
        if info:
            self._planes['mse'] = torch.zeros((steps, ld), dtype=f32, device=dev)
            self._planes['truncated'] = torch.zeros((steps, ld), dtype=torch.bool, device=dev)

    def sink_tensors(self):
        return self._planes

    @property
    def obs(self):
        return self._planes['obs'][:, :, :self._n].permute(0, 2, 1)

    @property
    def obs_planes(self):
        """[T, obs_dim, ld]: the SoA planes themselves (slab k is a valid ``state`` / observation buffer of the engine)."""
        return self._planes['obs']

    @property
    def reward(self):
        return self._planes['reward'][:, :self._n]

    @property
    def done(self):
        return self._planes['done'][:, :self._n]

    @property
    def constraint_values(self):
        v = self._planes['constraint_values'][:, :, :self._n].permute(0, 2, 1)
        if self._fold is None:
            return v
        t, n, r = v.shape                        # symmetric constraints: |y| - bound = max of the row pair (a copy)
        return self._fold(v.reshape(t * n, r)).reshape(t, n, -1)

    @property
    def constraint_violation(self):
        return self._planes['constraint_violation'][:, :self._n]

    @property
    def mse(self):
        return self._planes['mse'][:, :self._n]

    @property
    def truncated(self):
        return self._planes['truncated'][:, :self._n]


class BenchmarkEnv:
    """Batched base class.  Subclasses provide the system-specific pieces (spaces, constants)."""
    _count = 0
    NAME = 'base'
    AVAILABLE_CONSTRAINTS = None
    DISTURBANCE_MODES = None
    INERTIAL_PROP_RAND_INFO = None
    INIT_STATE_RAND_INFO = None
    TASK_INFO = None
    SYSTEM = None               # scg_system id

    def __init__(self,
                 output_dir=None, seed=None, info_in_reset=False, gui=False, verbose=False,
                 normalized_rl_action_space=False,
                 task=Task.STABILIZATION, task_info=None, cost=Cost.RL_REWARD,
                 pyb_freq=50, ctrl_freq=50, episode_len_sec=5,
                 init_state=None, randomized_init=True, init_state_randomization_info=None,
                 prior_prop=None, inertial_prop=None, randomized_inertial_prop=False,
                 inertial_prop_randomization_info=None,
                 constraints=None, done_on_violation=False, use_constraint_penalty=False, constraint_penalty=-1,
                 disturbances=None, adversary_disturbance=None, adversary_disturbance_offset=0.0,
                 adversary_disturbance_scale=0.01,
                 # ---- batched-engine arguments (new) ----
                 num_envs=1, device='cuda', physics='rk4', integration_substeps=1, motor_map=True,
                 auto_reset=True, info_level='full', compensated_state=False, track_episode_stats=True,
                 env_index_base=0, rollout_queues=None, rollout_split_min=None, ref_in_lds=None, grid_cap_per_cu=None, rollout_graph=None,
                 plane_pad=None, copy_outputs=None, lone_launch_forms=None,
                 **kwargs):
        self.id = self.__class__._count
        self.__class__._count += 1
        self.output_dir = os.getcwd() if output_dir is None else output_dir
        self.GUI, self.VERBOSE = gui, verbose
        if gui:
            raise ValueError('[ERROR] the batched engine has no GUI (gui=True is not supported).')
        self.num_envs = int(num_envs)
        if self.num_envs < 1:
            raise ValueError('[ERROR] num_envs must be >= 1.')
        self.device = torch.device(device)
        self.TASK = Task(task)
        if task_info is not None:
            self.TASK_INFO = task_info
        # timing constants (benchmark_env.py:151-161)
        self.CTRL_FREQ, self.PYB_FREQ = ctrl_freq, pyb_freq
        if self.PYB_FREQ % self.CTRL_FREQ != 0:
            raise ValueError('[ERROR] in BenchmarkEnv.__init__(), pyb_freq is not divisible by env_freq.')
        self.PYB_STEPS_PER_CTRL = int(self.PYB_FREQ / self.CTRL_FREQ)
        self.CTRL_TIMESTEP = 1. / self.CTRL_FREQ
        self.PYB_TIMESTEP = 1. / self.PYB_FREQ
        self.EPISODE_LEN_SEC = episode_len_sec
        self.CTRL_STEPS = self.EPISODE_LEN_SEC * self.CTRL_FREQ
        # integrator selection
        phys = str(getattr(physics, 'value', physics)).lower()
        if phys not in _PHYSICS_ALIASES:
            raise ValueError('[ERROR] unknown physics %r (use rk4 | euler, or a reference Physics value).' % physics)
        self.PHYSICS = phys
        self.INTEGRATOR = _PHYSICS_ALIASES[phys]
        if integration_substeps == 'pyb':
            integration_substeps = self.PYB_STEPS_PER_CTRL
        self.INTEGRATION_SUBSTEPS = int(integration_substeps)
        self.MOTOR_MAP = bool(motor_map)
        self.AUTO_RESET = bool(auto_reset)
        if info_level not in ('full', 'minimal'):
            raise ValueError("[ERROR] info_level must be 'full' or 'minimal'.")
        self.INFO_LEVEL = info_level
        # step() outputs: fresh tensors (the reference returns new arrays, quadrotor.py:878 deepcopy) or zero-copy views of
        # the engine's planes that the NEXT step() overwrites.  Default: fresh with the drop-in API (info_level='full'),
        # views with the lean one.  See step().
        self.COPY_OUTPUTS = (info_level == 'full') if copy_outputs is None else bool(copy_outputs)
        # how fresh outputs are produced: the kernel writes this step's obs / rew / done straight into newly allocated tensors
        # (an obs buffer apart from the state planes: +4 obs_dim bytes of stores per env-step, no copy kernel); the precise
        # modes, whose kernels are not built with a separate obs buffer, clone the views instead
        self._fresh_outputs = False          # decided in _alloc_buffers
        self._handed_out = False             # fresh-output mode: a tensor of the engine's buffer table went out through step() / reset()
        self._pingpong = False
        # precise mode: state kept as hi + lo planes and accumulated in float64.  True / 'full': every state,
        # float32 residual planes, float64 arithmetic throughout; 'attitude' (3-D quadrotor): residual planes and double
        # accumulation for the rotational chain only (attitude + body rates), float32 positions and velocities, the residuals as
        # int16 planes in units of 2^-16 ulp(hi) (40 significant bits) -- 72 B per env-step less than 'full', same <= 1e-5 drift
        # bound; 'attitude_f32': the same with float32 residual planes (48 B less than 'full'; scg_step_io.state_lo_kind = 1)
        if compensated_state in (True, 'full'):
            self.COMPENSATED_STATE = 'full'
        elif compensated_state in ('attitude', 'attitude_f32'):
            self.COMPENSATED_STATE = 'attitude'
        elif not compensated_state:
            self.COMPENSATED_STATE = False
        else:
            raise ValueError("[ERROR] compensated_state must be False, True / 'full', 'attitude' or 'attitude_f32'.")
        self.RESIDUAL_DTYPE = torch.float32 if compensated_state == 'attitude_f32' or self.COMPENSATED_STATE == 'full' else torch.int16
        self.TRACK_EPISODE_STATS = bool(track_episode_stats)
        self.ENV_INDEX_BASE = int(env_index_base)
        # launch policy of this env's handle (scg_config fields; the SCG_* environment variables are only the defaults
        # of these keyword arguments, read here -- the library itself reads no environment variable)
        def _knob(value, env_name, default):
            return int(os.environ.get(env_name, default)) if value is None else int(value)
        self.ROLLOUT_QUEUES = _knob(rollout_queues, 'SCG_ROLLOUT_QUEUES', 0)        # 0 = library default (2)
        self.ROLLOUT_SPLIT_MIN = _knob(rollout_split_min, 'SCG_ROLLOUT_SPLIT_MIN', 0)
        self.REF_IN_LDS = _knob(ref_in_lds, 'SCG_REF_IN_LDS', 0)
        self.GRID_CAP_PER_CU = _knob(grid_cap_per_cu, 'SCG_GRID_CAP_PER_CU', 0)
        self.ROLLOUT_GRAPH = _knob(rollout_graph, 'SCG_ROLLOUT_GRAPH', 0)           # 0 / 1: hipGraph replay of small-batch rollouts; -1 off
        self.PLANE_PAD = _knob(plane_pad, 'SCG_LD_PAD', 17)                         # units of 64 lanes (256 B)
        self.LONE_LAUNCH_FORMS = _knob(lone_launch_forms, 'SCG_LONE_LAUNCH_FORMS', 0)   # 0 / 1: step() launches use the lone-launch kernel forms; -1 off
        # initial state / randomisation
        self.INIT_STATE = init_state
        self.RANDOMIZED_INIT = randomized_init
        if init_state_randomization_info is not None:
            self.INIT_STATE_RAND_INFO = init_state_randomization_info
        self.PRIOR_PROP = prior_prop
        self.INERTIAL_PROP = inertial_prop
        self.RANDOMIZED_INERTIAL_PROP = randomized_inertial_prop
        if inertial_prop_randomization_info is not None:
            self.INERTIAL_PROP_RAND_INFO = inertial_prop_randomization_info
        self.NORMALIZED_RL_ACTION_SPACE = normalized_rl_action_space
        self.COST = Cost(cost)
        if self.COST == Cost.COMPETITION:
            raise NotImplementedError('[ERROR] cost=competition needs PyBullet contact/ray queries (out of scope).')
        self._set_action_space()
        self._set_observation_space()
        self.action_dim = self.action_space.shape[0]
        self.obs_dim = self.observation_space.shape[0]
        self.state_dim = self.state_space.shape[0] if hasattr(self, 'state_space') else self.obs_dim
        if self.COST == Cost.QUADRATIC:
            self.Q = np.eye(self.state_dim)          # benchmark_env.py:190-192 (obs == state for quadratic cost)
            self.R = np.eye(self.action_dim)
        self.CONSTRAINTS = constraints
        self.DONE_ON_VIOLATION = done_on_violation
        self.use_constraint_penalty = use_constraint_penalty
        self.constraint_penalty = constraint_penalty
        self._engine = None
        self._setup_constraints()
        self.DISTURBANCES = disturbances
        self.adversary_disturbance = adversary_disturbance
        self.adversary_disturbance_offset = adversary_disturbance_offset
        self.adversary_disturbance_scale = adversary_disturbance_scale
        self._setup_disturbances()
        self.seed(seed)
        self.initial_reset = False
        self.INFO_IN_RESET = info_in_reset
        self.RND_SEED = seed
        self.RESEED_ON_RESET = kwargs.get('reseed_on_reset', False)
        self._bufs = None
        self._raw_action, self._raw_action_rows = None, False
        self._async_outs, self._inflight = None, None            # step_async() / step_wait()
        self._state_external, self._bound_dirty = False, False   # step(out=...) in views mode
        self._capture_base, self._serial_word = None, None       # capture_loop(): the steps being recorded carry serial OFFSETS
        # output_ring=K >= 3 (fresh-output mode): step() cycles through K pre-allocated output sets instead of allocating one per
        # step -- a tensor handed out is then OVERWRITTEN K steps later (opt-in: it changes the ownership contract of step())
        self._output_ring = int(kwargs.get('output_ring', 0) or 0)
        if self._output_ring and self._output_ring < 3:
            raise ValueError('[ERROR] output_ring must be 0 (off) or >= 3.')
        self._ring, self._ring_pos = None, 0
        # output_pool (fresh-output mode without a ring; default 'auto'): step() RECYCLES an output set -- planes, reward, done and the
        # views of them it hands out, all made once -- as soon as the caller holds no reference to any of it any more (Python
        # reference counts of the objects handed out + the use count of their storage: views the caller derived count too), instead
        # of three torch.empty and three tensor views per step (6-7 us of host time: half the cost of a default-tier step at the
        # reference's batch sizes).  The ownership contract is unchanged: a tensor, once returned, is never written again while
        # anybody can still read it.  An int caps the number of sets (>= 4), 0 switches it off; 'auto' = up to 8 sets within 128 MiB.
        self._output_pool = kwargs.get('output_pool', 'auto')
        if self._output_pool not in ('auto', None) and (not isinstance(self._output_pool, int) or self._output_pool < 0
                                                         or 0 < self._output_pool < 4):
            raise ValueError("[ERROR] output_pool must be 'auto', 0 (off) or an int >= 4.")
        self._pool = None
        # terminal_layout='planes': info['terminal_observation'] / info['episode'] live in SoA planes [obs_dim + 2][ld] instead of one
        # dense record per env -- for a caller that runs a critic over the terminal observations of ALL envs every step (the
        # time-limit bootstrap, controllers/ppo/ppo.py:264-273): it then reads 4 obs_dim bytes per env, coalesced, instead of the
        # records' stride; the envs that finish pay obs_dim + 2 small stores instead of one record
        self._terminal_planes = {'records': False, 'planes': True}.get(kwargs.get('terminal_layout', 'records'))
        if self._terminal_planes is None:
            raise ValueError("[ERROR] terminal_layout must be 'records' or 'planes'.")

    # ------------------------------------------------------------------ seeding
    def seed(self, seed=None):
        """benchmark_env.py:220-235.  The device stream is Philox keyed by this seed; ``None`` draws one."""
        if seed is None:
            seed = int(np.random.SeedSequence().entropy % (2 ** 63))
        # the Philox key is part of the handle's constants: a new handle only when the key really changes
        if getattr(self, '_seed', None) != int(seed):
            self._engine_dirty = True
        self._seed = int(seed)
        self.np_random = np.random.RandomState(self._seed % (2 ** 32))
        self.action_space.seed(self._seed % (2 ** 32))
        for disturbs in getattr(self, 'disturbances', {}).values():
            disturbs.seed(self)
        # re-seeding rewinds the stream: the draws that follow repeat those that followed the previous seed() with this value
        # (counter = (global env index, serial); the reference re-creates np_random, benchmark_env.py:231)
        self._serial = 0
        return [seed]

    # -- checkpoint / resume of the simulation state (the reference checkpoints only the policy side and the envs' RNG
    #    states, controllers/ppo/ppo.py:115-155 + dummy_vec_env.py:88-100; here the env state is a handful of tensors) ----
    # 'obs' is persistent too when it is a buffer of its own (goal horizon / observation noise): the observation an env
    # exposes between two steps must survive a resume
    _CHECKPOINT_PLANES = ('state', 'state_lo', 'step_count', 'ep_return', 'ep_length', 'inertial', 'episode_id', 'obs')

    def state_dict(self):
        """Everything a resumed run needs to continue bit for bit: the device planes that persist across steps, the
        position of the Philox streams (seed, step serial) and the per-env nominal reset states.  ``torch.save``-able."""
        self._refuse_while_pending('state_dict()')
        self._ensure_engine()
        return {'num_envs': self.num_envs, 'plane_stride': self._ld, 'seed': self._seed, 'serial': self._serial,
                'initial_reset': self.initial_reset,
                'planes': {k: self._bufs[k].clone() for k in self._CHECKPOINT_PLANES if k in self._bufs},
                'reset_state': None if self._reset_state_buf is None else self._reset_state_buf.clone()}

    # -- recording the caller's step loop in a HIP graph (envs/graph_loop.py::capture_loop) ---------------------------------------
    def capture_loop(self, body, steps, warmup=2):
        """``steps`` iterations of ``body(k)`` -- capture-safe torch code around this env's ``step`` -- recorded in one HIP graph;
        ``.replay()`` runs them with one launch and equals eager stepping bit for bit (``envs/graph_loop.py``)."""
        from safe_control_gym_amd.envs.graph_loop import capture_loop
        return capture_loop([self], body, steps, warmup)

    def _begin_capture(self):
        self._ensure_engine()
        self._check_initial_reset()
        self._refuse_while_pending('capture_loop()')
        if self._fresh_outputs or (self.COPY_OUTPUTS and self.COMPENSATED_STATE):
            raise ValueError('[ERROR] capture_loop() needs an env that steps in place (copy_outputs=False): a replayed graph writes the '
                             'same buffers every time, fresh per-step output tensors cannot exist in it.')
        if self.adversary_disturbance is not None:
            raise ValueError('[ERROR] capture_loop(): the adversary acts through host code every step (set_adversary_control); not recorded.')
        if self._serial_word is None:
            self._serial_word = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._engine.set_serial_word(self._serial_word)
        self._capture_base = self._serial

    def _end_capture(self):
        """-> the number of step serials the recorded stretch consumes per replay.  Nothing ran: the serial goes back."""
        used = self._serial - self._capture_base
        self._serial, self._capture_base = self._capture_base, None
        self._engine.set_serial_word(None)
        return used

    def _before_replay(self):
        v = self._serial & 0xffffffff
        self._serial_word.fill_(v - (1 << 32) if v >= (1 << 31) else v)

    def _launch_serial(self):
        """The serial a launch carries: the env's running serial, or -- while a loop is being recorded -- its offset from the
        serial the recording started at (the kernel adds the serial word, written before every replay)."""
        return self._serial if self._capture_base is None else self._serial - self._capture_base

    def _refuse_while_pending(self, what):
        """A step issued by step_async() runs on the env's own queue until step_wait() joins it: nothing else may touch the planes."""
        if self._async_outs is not None:
            raise RuntimeError('[ERROR] %s while a step_async() is pending: call step_wait() first.' % what)

    def load_state_dict(self, sd):
        self._refuse_while_pending('load_state_dict()')
        if sd['num_envs'] != self.num_envs:
            raise ValueError('[ERROR] checkpoint holds %d envs, this env has %d.' % (sd['num_envs'], self.num_envs))
        if sd['seed'] != self._seed:
            self.seed(sd['seed'])
        self._ensure_engine()
        if sd['plane_stride'] != self._ld or set(sd['planes']) != {k for k in self._CHECKPOINT_PLANES if k in self._bufs}:
            raise ValueError('[ERROR] checkpoint was taken from a differently configured env.')
        for k, v in sd['planes'].items():
            # copy_() would cast silently: float32 residual planes ('attitude_f32', or a checkpoint of an earlier version) loaded
            # into an int16 'attitude' env would truncate every residual to 0, and the other way round turn integer counts
            # into radians
            if v.dtype != self._bufs[k].dtype or tuple(v.shape) != tuple(self._bufs[k].shape):
                raise ValueError('[ERROR] checkpoint was taken from a differently configured env: plane %r is %s %s there, %s %s here.'
                                 % (k, v.dtype, tuple(v.shape), self._bufs[k].dtype, tuple(self._bufs[k].shape)))
        self._own_state()
        for k, v in sd['planes'].items():
            self._bufs[k].copy_(v)
        self._reset_state_buf = None if sd['reset_state'] is None else sd['reset_state'].to(self.device).clone()
        self._serial = int(sd['serial'])
        self.initial_reset = bool(sd['initial_reset'])
        self._bound = False
        self._raw_action, self._raw_action_rows = None, False

    def set_cost_function_param(self, Q, R):
        """benchmark_env.py:237-254."""
        if not self.initial_reset:
            self.Q = np.asarray(Q, dtype=np.float64)
            self.R = np.asarray(R, dtype=np.float64)
            self._engine_dirty = True
        else:
            raise RuntimeError('[ERROR] env.set_cost_function_param() cannot be called after the first reset of the environment.')

    def set_adversary_control(self, action):
        """benchmark_env.py:256-266."""
        if self.adversary_disturbance is not None:
            # kept raw: clip(a, -1, 1) * scale + offset (benchmark_env.py:262-264) is formed while the planes are filled -- two kernels
            # over the batch instead of four (clamp, scale, offset, transposing copy: 25 us of every step at 2^20 envs)
            self._adv_raw = torch.as_tensor(action, dtype=torch.float32, device=self.device).reshape(self.num_envs, -1)
            self._adv_fresh = True
        else:
            raise RuntimeError('[ERROR] adversary_disturbance does not exist, env.set_adversary_control() cannot be called.')

    def _check_initial_reset(self):
        if not self.initial_reset:
            raise RuntimeError('[ERROR] You must call env.reset() at least once before using env.step().')

    # ------------------------------------------------------------------ setup hooks
    def _setup_disturbances(self):
        """benchmark_env.py:314-330."""
        from safe_control_gym_amd.envs.disturbances import create_disturbance_list
        self.disturbances = {}
        if self.DISTURBANCES is not None:
            for mode, specs in self.DISTURBANCES.items():
                assert mode in self.DISTURBANCE_MODES, '[ERROR] in BenchmarkEnv._setup_disturbances(), disturbance mode not available.'
                self.disturbances[mode] = create_disturbance_list(specs, self.DISTURBANCE_MODES[mode], self)
        self._adv_raw, self._adv_fresh, self._adv_force_live = None, False, False
        if self.adversary_disturbance is not None:
            assert self.adversary_disturbance in self.DISTURBANCE_MODES, '[ERROR] in _setup_disturbances()'
            from safe_control_gym_amd.envs.spaces import Box
            dim = self.DISTURBANCE_MODES[self.adversary_disturbance]['dim']
            self.adversary_action_space = Box(low=-np.ones(dim), high=np.ones(dim))
            self.adversary_observation_space = self.observation_space

    def _setup_constraints(self):
        """benchmark_env.py:332-338."""
        self.constraints = None
        self.num_constraints = 0
        if self.CONSTRAINTS is not None:
            self.constraints = create_constraint_list(self.CONSTRAINTS, self.AVAILABLE_CONSTRAINTS, self)
            self.num_constraints = self.constraints.num_constraints

    def _set_action_space(self):
        raise NotImplementedError

    def _set_observation_space(self):
        raise NotImplementedError

    def _fill_system_config(self, cfg):
        """Subclass: physical constants, action bounds, goals, out-of-bound box, init state."""
        raise NotImplementedError

    # ------------------------------------------------------------------ engine
    def _x_goal_table(self):
        return np.asarray(self.X_GOAL, dtype=np.float64) if self.TASK == Task.TRAJ_TRACKING else None

    def _make_engine(self, constraint_list=None):
        cfg = _capi.ScgConfig()
        cfg.struct_size = _capi.C.sizeof(_capi.ScgConfig)
        cfg.abi_version = _capi.SCG_ABI_VERSION
        cfg.system = self.SYSTEM
        cfg.integrator = _INTEGRATORS[self.INTEGRATOR]
        cfg.substeps = self.INTEGRATION_SUBSTEPS
        cfg.task = _capi.SCG_TRAJ_TRACKING if self.TASK == Task.TRAJ_TRACKING else _capi.SCG_STABILIZATION
        cfg.cost = _capi.SCG_QUADRATIC if self.COST == Cost.QUADRATIC else _capi.SCG_RL_REWARD
        cfg.ctrl_steps = int(self.CTRL_STEPS)
        cfg.dt = self.CTRL_TIMESTEP
        cfg.normalized_action = int(bool(self.NORMALIZED_RL_ACTION_SPACE))
        cfg.motor_map = int(self.MOTOR_MAP)
        nx, nu = self.state_dim, self.action_dim
        for j in range(nu):
            cfg.act_low[j] = float(self.action_space.low[j])
            cfg.act_high[j] = float(self.action_space.high[j])
            cfg.u_goal[j] = float(self.U_GOAL[j])
        if self.TASK == Task.STABILIZATION:
            for j in range(nx):
                cfg.x_goal[j] = float(self.X_GOAL[j])
        sw = np.broadcast_to(np.asarray(self.rew_state_weight, dtype=np.float64), (nx,))
        aw = np.broadcast_to(np.asarray(self.rew_act_weight, dtype=np.float64), (nu,))
        for j in range(nx):
            cfg.rew_state_weight[j] = float(sw[j])
            cfg.mse_weight[j] = float(self.info_mse_metric_state_weight[j])
        for j in range(nu):
            cfg.rew_act_weight[j] = float(aw[j])
        cfg.rew_exponential = int(bool(self.rew_exponential))
        if self.COST == Cost.QUADRATIC:
            Q = np.asarray(self.Q, dtype=np.float64).reshape(nx, nx)
            R = np.asarray(self.R, dtype=np.float64).reshape(nu, nu)
            for a in range(nx):
                for b in range(nx):
                    cfg.Q[a * _capi.SCG_MAX_NX + b] = float(Q[a, b])
            for a in range(nu):
                for b in range(nu):
                    cfg.R[a * _capi.SCG_MAX_NU + b] = float(R[a, b])
        cfg.done_on_out_of_bound = int(bool(self.done_on_out_of_bound))
        cfg.done_on_violation = int(bool(self.DONE_ON_VIOLATION))
        cfg.goal_tolerance = float(self.TASK_INFO['stabilization_goal_tolerance'])
        cfg.use_constraint_penalty = int(bool(self.use_constraint_penalty))
        cfg.constraint_penalty = float(self.constraint_penalty)
        cfg.obs_goal_horizon = int(self.obs_goal_horizon)
        cfg.randomized_init = int(bool(self.RANDOMIZED_INIT))
        cfg.auto_reset = int(self.AUTO_RESET)
        cfg.randomized_inertial = int(bool(self.RANDOMIZED_INERTIAL_PROP))
        if self.RANDOMIZED_INERTIAL_PROP:
            for q, key in enumerate(self._INERTIAL_KEYS):
                info = self.INERTIAL_PROP_RAND_INFO.get(key) if key is not None else None
                if info is None:
                    continue
                from safe_control_gym_amd.envs.randomization import parse_distribution, lower_bound
                kind, lo, hi, prm = parse_distribution(info, key)                  # benchmark_env.py:297-307
                # quadrotor.py:464-465 raises when a DRAW comes out negative.  A batched engine cannot abort one env, so: a
                # distribution whose support reaches a non-positive property is refused here; an unbounded one (normal,
                # laplace, logistic) is accepted and the kernel floors the property at 0.1 % of its nominal value.
                if self._inertial_nominal()[q] + lower_bound(kind, lo, prm) <= 0 and lower_bound(kind, lo, prm) > float('-inf'):
                    raise ValueError('[ERROR] in reset(), negative randomized inertial properties.')
                cfg.inertial_rand_kind[q] = kind
                cfg.inertial_rand_low[q], cfg.inertial_rand_high[q] = lo, hi
                for r in range(3):
                    cfg.inertial_rand_p[q][r] = prm[r]
        cfg.seed = self._seed & 0xFFFFFFFFFFFFFFFF
        cfg.env_index_base = self.ENV_INDEX_BASE
        cfg.rollout_queues = self.ROLLOUT_QUEUES
        cfg.rollout_split_min = self.ROLLOUT_SPLIT_MIN
        cfg.ref_in_lds = self.REF_IN_LDS
        cfg.grid_cap_per_cu = self.GRID_CAP_PER_CU
        cfg.rollout_graph = self.ROLLOUT_GRAPH
        cfg.lone_launch_forms = self.LONE_LAUNCH_FORMS
        self._fill_system_config(cfg)
        return Engine(cfg, x_goal_table=self._x_goal_table(), constraint_list=constraint_list, device=self.device,
                      disturbances=[e for mode, dl in self.disturbances.items() for e in dl.compile(mode)])

    def _ensure_engine(self):
        if self._engine is None or self._engine_dirty:
            if self._engine is not None:
                self._engine.close()
            self._engine = self._make_engine(constraint_list=self.constraints)
            self._engine_dirty = False
        if self._bufs is None:
            self._alloc_buffers()

    def _alloc_buffers(self):
        """SoA planes, padded to a multiple of 64 lanes so every plane starts 256-B aligned."""
        n = self.num_envs
        ld = (n + 63) // 64 * 64
        # Plane-stride skew: with a power-of-two stride the j-th plane of every array maps to the same HBM
        # channel for a given env index and the 12 + 4 + ... streams of a wave collide there.  17 x 256 B of
        # padding per plane measured best (-10 % kernel time at 2^20 envs; 64 x 256 B is as bad as none) --
        # DESIGN.md section 6.  make(..., plane_pad=k) (units of 64 lanes) overrides it for tuning.
        if n >= 4096:
            ld += 64 * self.PLANE_PAD
        self._ld = ld
        dev, f32 = self.device, torch.float32
        nx, nu = self.state_dim, self.action_dim
        b = {}
        b['state'] = torch.zeros((nx, ld), dtype=f32, device=dev)
        b['action'] = torch.zeros((nu, ld), dtype=f32, device=dev)
        b['step_count'] = torch.zeros(ld, dtype=torch.int32, device=dev)
        b['reward'] = torch.zeros(ld, dtype=f32, device=dev)
        b['done'] = torch.zeros(ld, dtype=torch.bool, device=dev)          # the kernel writes 0 / 1 bytes
        if any(dl.needs_episode_id for dl in self.disturbances.values()):
            b['episode_id'] = torch.zeros(ld, dtype=torch.int32, device=dev)     # uint32 on the device side
        if self.COMPENSATED_STATE == 'attitude':
            if self.SYSTEM != _capi.SCG_QUAD_3D:
                raise ValueError("[ERROR] compensated_state='attitude' is defined for the 3-D quadrotor (use True elsewhere).")
            b['state_lo'] = torch.zeros((6, ld), dtype=self.RESIDUAL_DTYPE, device=dev)        # residuals of states 6..11 (int16: units of 2^-16 ulp)
        elif self.COMPENSATED_STATE:
            b['state_lo'] = torch.zeros((nx, ld), dtype=f32, device=dev)
        if self.RANDOMIZED_INERTIAL_PROP:
            b['inertial'] = torch.zeros((4, ld), dtype=f32, device=dev)
        if self.TRACK_EPISODE_STATS:
            b['ep_return'] = torch.zeros(ld, dtype=f32, device=dev)
            if not self.AUTO_RESET:          # with auto-reset the episode length IS the step counter: no second plane
                b['ep_length'] = torch.zeros(ld, dtype=torch.int32, device=dev)
        if self.constraints is not None:
            b['constraint_values'] = torch.zeros((self.constraints.num_device_rows, ld), dtype=f32, device=dev)
            b['constraint_violation'] = torch.zeros(ld, dtype=torch.uint8, device=dev)
        self._fresh_outputs = self.COPY_OUTPUTS and not self.COMPENSATED_STATE
        # fresh outputs of a noise-free observation: the state ping-pongs.  Every step writes the new state (scg_step_io.state_out)
        # into a newly allocated buffer, which IS the observation handed out and the next step's input: no byte is moved twice
        self._pingpong = self._fresh_outputs and 'observation' not in self.disturbances
        if self.obs_dim != nx and 'observation' not in self.disturbances and (self._pingpong or not self.COPY_OUTPUTS):
            # noise-free goal-horizon observation: one buffer of obs_dim planes whose head IS the state (scg_step_io.obs ==
            # state): the kernel writes the state once and only the goal rows behind it
            b['obs'] = torch.zeros((self.obs_dim, ld), dtype=f32, device=dev)
            b['state'] = b['obs'][:nx]
        elif self.obs_dim != nx or 'observation' in self.disturbances:
            b['obs'] = torch.zeros((self.obs_dim, ld), dtype=f32, device=dev)
        if self.INFO_LEVEL == 'full':
            b['mse'] = torch.zeros(ld, dtype=f32, device=dev)
            b['truncated'] = torch.zeros(ld, dtype=torch.bool, device=dev)
            # one dense row per env: terminal observation | finished-episode return | length (int32 bits), written by
            # the lanes that finish an episode (a multiple of 8 floats: whole 32-byte DRAM sectors)
            b['terminal_record'] = torch.zeros((self.obs_dim + 2, ld) if self._terminal_planes else (ld, (self.obs_dim + 2 + 7) // 8 * 8),
                                               dtype=f32, device=dev)
            if self.TASK == Task.STABILIZATION and self.COST == Cost.QUADRATIC:
                b['goal_reached'] = torch.zeros(ld, dtype=torch.bool, device=dev)
        # the adversary's planes (set_adversary_control): static buffers, so the step keeps its bound fast path (and its own kernel
        # tier, FT_ADV); zero until the adversary acts
        if self.adversary_disturbance == 'action':
            b['action_disturbance'] = torch.zeros((self.action_dim, ld), dtype=f32, device=dev)
        elif self.adversary_disturbance == 'dynamics':
            b['dynamics_force'] = torch.zeros((3, ld), dtype=f32, device=dev)
        self._adv_force_live = False
        self._bufs = b
        self._reset_state_buf = None
        self._bound = False
        self._ring, self._ring_pos = None, 0
        if self._output_ring and self._fresh_outputs:
            # per slot: planes, reward, done, and -- made once, a tensor view costs 1-2 us of host time -- the state rows of the
            # planes and the (obs, rew, done, planes) views step() hands out
            self._ring = []
            for _ in range(self._output_ring):
                planes = torch.zeros((self.obs_dim, ld), dtype=f32, device=dev)
                rew_t, done_t = torch.zeros(n, dtype=f32, device=dev), torch.zeros(n, dtype=torch.bool, device=dev)
                self._ring.append((planes, rew_t, done_t, planes if self.obs_dim == nx else planes[:nx],
                                   (planes[:, :n].T, rew_t, done_t, planes)))
        self._pool = None
        if self._fresh_outputs and self._ring is None and self._output_pool not in (0, None):
            slot_bytes = self.obs_dim * ld * 4 + 5 * n + 512
            cap = min(8, (128 << 20) // slot_bytes) if self._output_pool == 'auto' else int(self._output_pool)
            if cap >= 4:
                self._pool = RecyclingPool(functools.partial(self._new_output_set, n, ld, self.obs_dim, nx, dev), cap)
        # output views are created once: step() hands out the same zero-copy views of the buffers every call
        info = {}
        if 'mse' in b:
            info['mse'] = b['mse'][:n]
        if 'goal_reached' in b:
            info['goal_reached'] = b['goal_reached'][:n]
        if self.constraints is not None:
            info['constraint_values'] = self.constraints.fold(b['constraint_values'][:, :n].T) \
                if self.constraints.num_device_rows == self.constraints.num_constraints else None
            info['constraint_violation'] = b['constraint_violation'][:n]
        if 'truncated' in b:
            info['TimeLimit.truncated'] = b['truncated'][:n]
        if 'terminal_record' in b:
            rec, od = b['terminal_record'], self.obs_dim
            if self._terminal_planes:
                info['terminal_observation'] = rec[:od, :n].T                  # [N, obs_dim] zero-copy transpose, as the observation is
                if self.TRACK_EPISODE_STATS:
                    info['episode'] = {'r': rec[od, :n], 'l': rec.view(torch.int32)[od + 1, :n]}
            else:
                info['terminal_observation'] = rec[:n, :od]
                if self.TRACK_EPISODE_STATS:
                    info['episode'] = {'r': rec[:n, od], 'l': rec.view(torch.int32)[:n, od + 1]}
        self._info_views = info
        self._out_views = ((b['obs'] if 'obs' in b else b['state'])[:, :n].T, b['reward'][:n], b['done'][:n])
        self._out_view_cache = {}

    # ------------------------------------------------------------------ views
    @property
    def state(self):
        """[N, nx] view of the SoA state planes (zero copy); None before the first reset(), as in the reference."""
        if self._bufs is None:
            return None
        return self._bufs['state'][:, :self.num_envs].T

    @property
    def state_soa(self):
        return self._bufs['state'][:, :self.num_envs]

    @property
    def residual_planes(self):
        """The residual planes of a precise mode ([nx | 6, ld]; float32, or int16 in the attitude mode), None otherwise."""
        self._ensure_engine()
        return self._bufs.get('state_lo')

    @property
    def state_precise(self):
        """[N, nx] float64: the state with its residual planes added (precise modes; the float32 state otherwise).  int16
        residuals count 2^(e - 40) where the float32 plane value is f 2^e, f in [1/2, 1) (scg_step_io.state_lo_kind = 2)."""
        x = self.state.double().clone()
        lo = self._bufs.get('state_lo') if self._bufs is not None else None
        if lo is None:
            return x
        lo = lo[:, :self.num_envs].T
        first = self.state_dim - lo.shape[1]
        if lo.dtype == torch.int16:
            e = torch.frexp(self.state[:, first:])[1]
            x[:, first:] += torch.ldexp(lo.double(), (e - 40))
        else:
            x[:, first:] += lo.double()
        return x

    @property
    def action_buffer(self):
        """[N, nu] view of the SoA action planes: write actions here to skip the transpose copy."""
        return self._bufs['action'][:, :self.num_envs].T

    @property
    def ctrl_step_counter(self):
        """[N] int32 view of the per-env step counter (benchmark_env.py:366,428); 0 before the first reset()."""
        if self._bufs is None:
            return 0
        return self._bufs['step_count'][:self.num_envs]

    @property
    def pyb_step_counter(self):
        return self.ctrl_step_counter * self.PYB_STEPS_PER_CTRL

    @property
    def current_preprocessed_action(self):
        """[N, nu] the action after ``_preprocess_control`` (quadrotor.py:815-833: clipped, or mapped from the normalised space),
        recomputed from the raw action on demand (the kernel does not store it); None before the first step()."""
        raw = self.current_raw_input_action
        if raw is None:
            return None
        lo = torch.as_tensor(self.action_space.low, device=raw.device)
        hi = torch.as_tensor(self.action_space.high, device=raw.device)
        a = torch.nan_to_num(raw, nan=0.0).clamp(lo, hi)
        a = torch.where(torch.isnan(raw), lo.expand_as(raw), a)             # the kernel reads a NaN as the lower bound
        if getattr(self, 'NORMALIZED_RL_ACTION_SPACE', False):
            if self.NAME == 'cartpole':
                return a * float(getattr(self, 'action_scale', 1.0))
            return (1.0 + float(self.norm_act_scale) * a) * float(self.hover_thrust)
        return a

    @property
    def cnstr_violation(self):
        """benchmark_env.py:218,441-449: whether the last step violated a constraint -- [N] bool per env (False without
        constraints or before the first step)."""
        if self._bufs is None or 'constraint_violation' not in self._bufs:
            return False
        return self._bufs['constraint_violation'][:self.num_envs].bool()

    @property
    def goal_reached(self):
        """quadrotor.py:965: [N] bool of the last step (stabilisation with the quadratic cost, info_level='full'); else False."""
        if self._bufs is None or 'goal_reached' not in self._bufs:
            return False
        return self._bufs['goal_reached'][:self.num_envs]

    @property
    def inertial_properties(self):
        """[N, 4] per-env (M, Ixx, Iyy, Izz) | (pole_length, pole_mass, cart_mass, -) when randomized_inertial_prop."""
        return self._bufs['inertial'][:, :self.num_envs].T

    @property
    def episode_return(self):
        return self._bufs['ep_return'][:self.num_envs]

    @property
    def episode_length(self):
        return self._bufs.get('ep_length', self._bufs['step_count'])[:self.num_envs]

    @property
    def current_raw_input_action(self):
        """[N, nu] view of the last raw action (benchmark_env.py:431 keeps it for the input constraints)."""
        if self._raw_action is None:
            return None
        return self._raw_action if self._raw_action_rows else self._raw_action[:, :self.num_envs].T

    def _raw_action_soa(self):
        """The last raw action as planes [nu, ld] (for the stand-alone constraint kernel)."""
        if self._raw_action is None:
            return self._bufs['action']
        if self._raw_action_rows:
            buf = self._bufs['action']
            buf[:, :self.num_envs].copy_(self._raw_action.T)
            return buf
        return self._raw_action

    # -- batched prior-model queries ---------------------------------------
    def _planes(self, t, dim, default=None):
        """(planes [dim, ld'], ld') from [N, dim] rows (one transposing copy), [dim, ld'] planes (zero copy),
        a [dim] vector (broadcast to every env) or None (-> ``default``)."""
        n = self.num_envs
        if t is None:
            t = default
        if isinstance(t, torch.Tensor) and t.ndim == 2 and t.shape[0] == dim and t.shape[1] >= n and t.shape != (n, dim) \
                and t.dtype == torch.float32 and t.is_contiguous() and t.device == self._bufs['state'].device:
            return t, t.shape[1]
        a = torch.as_tensor(np.asarray(t, dtype=np.float32) if not isinstance(t, torch.Tensor) else t,
                            dtype=torch.float32, device=self.device)
        if a.ndim == 1:
            a = a.reshape(1, dim).expand(n, dim)
        if a.shape != (n, dim):
            raise ValueError('[ERROR] expected (num_envs, %d) rows, (%d, ld) planes or a (%d,) vector, got %s'
                             % (dim, dim, dim, tuple(a.shape)))
        buf = torch.empty((dim, self._ld), dtype=torch.float32, device=self.device)
        buf[:, :n] = a.T
        return buf, self._ld

    def _model_planes(self, state, input):
        xs, ldx = self._planes(state, self.state_dim, default=self._bufs['state'])
        us, ldu = self._planes(input, self.action_dim, default=np.asarray(self.U_EQ, dtype=np.float64))
        if ldx != ldu:                                     # bring both to the env's own plane stride
            xs = self._planes(xs[:, :self.num_envs].T, self.state_dim)[0] if ldx != self._ld else xs
            us = self._planes(us[:, :self.num_envs].T, self.action_dim)[0] if ldu != self._ld else us
            ldx = self._ld
        inertial = self._bufs.get('inertial') if ldx == self._ld else None
        return xs, us, inertial, ldx

    def model_dynamics(self, state=None, input=None):
        """Batched ``symbolic.fc_func`` (symbolic_systems.py:68): ``x_dot = f(x, u)`` [N, nx] for each env's (state, input)
        (defaults as in ``linearize``)."""
        self._ensure_engine()
        xs, us, inertial, ld = self._model_planes(state, input)
        return self._engine.model_eval(xs, us, inertial, self.num_envs, ld, True, False)[0][:, :self.num_envs].T

    def model_step(self, state=None, input=None):
        """Batched ``symbolic.fd_func`` (symbolic_systems.py:70-73): the state after one control step of the prior model
        under zero-order-hold input, [N, nx], WITHOUT touching the envs -- the engine's own RK4 / Euler discretisation
        (the reference integrates with CVODES)."""
        self._ensure_engine()
        xs, us, inertial, ld = self._model_planes(state, input)
        return self._engine.model_eval(xs, us, inertial, self.num_envs, ld, False, True)[1][:, :self.num_envs].T

    def _rows64(self, t, dim, default):
        """[num_envs, dim] rows / a (dim,) vector / None -> contiguous float64 planes [dim, num_envs]."""
        n = self.num_envs
        if t is None:
            t = default
        a = t if isinstance(t, torch.Tensor) else torch.as_tensor(np.asarray(t, dtype=np.float64))
        a = a.to(device=self.device, dtype=torch.float64)
        if a.ndim == 1:
            a = a.reshape(1, dim).expand(n, dim)
        if a.shape != (n, dim):
            raise ValueError('[ERROR] expected (num_envs, %d) rows or a (%d,) vector, got %s' % (dim, dim, tuple(a.shape)))
        return a.T.contiguous()

    def model_flow(self, state=None, input=None, dt=None, rtol=1e-6, atol=1e-8, sensitivities=False, max_doublings=12,
                   return_intervals=False):
        """Batched ``symbolic.fd_func`` AS THE REFERENCE DEFINES IT (symbolic_systems.py:70-73: CasADi's CVODES integrator
        over ``[0, dt]``; its default tolerances are the defaults here): the state each env's prior model reaches after ``dt``
        (default ``CTRL_TIMESTEP``) under a constant model input, integrated in float64 to ``atol + rtol |x|`` (RK4 with
        step-doubling error control and order-5 extrapolation) -> ``xf`` [N, nx] float64.  ``sensitivities=True`` adds the
        integrator's forward sensitivities ``(xf, dxf_dx0 [N, nx, nx], dxf_dp [N, nx, nu])`` (~1e-9 relative).  ``state``
        defaults to the envs' current states, ``input`` to ``U_EQ``; rows may be float32 or float64.  ``model_step`` is the
        fixed-step RK4 discretisation of the same model (the reference's ``rk_discrete``)."""
        self._ensure_engine()
        n = self.num_envs
        xs = self._rows64(state, self.state_dim, self._bufs['state'][:self.state_dim, :n].T if state is None else None)
        us = self._rows64(input, self.action_dim, np.asarray(self.U_EQ, dtype=np.float64))
        inertial = self._bufs.get('inertial')
        if inertial is not None:
            inertial = inertial[:, :n].to(torch.float64).contiguous()
        xf, sx, su, intervals = self._engine.model_flow(xs, us, inertial, n, n, self.CTRL_TIMESTEP if dt is None else dt, rtol, atol,
                                                        want_sens=sensitivities, max_doublings=max_doublings)
        out = (xf.T, sx, su) if sensitivities else xf.T
        if return_intervals:
            return (out + (intervals,)) if sensitivities else (out, intervals)
        return out

    def linearize(self, state=None, input=None, discrete=False):
        """Batched ``symbolic.df_func`` (symbolic_systems.py:77-92) on the device: Jacobians of the continuous
        dynamics about each env's (state, input) -> ``(dfdx [N, nx, nx], dfdu [N, nx, nu])`` float32 tensors.

        ``state`` defaults to the envs' current states (zero copy), ``input`` (model input u: thrusts / force,
        i.e. the CLIPPED action) to ``U_EQ``.  ``discrete=True`` applies ``discretize_linear_system(exact=False)``
        (mpc_utils.py:51-54): ``Ad = I + A dt``, ``Bd = B dt`` with ``dt = CTRL_TIMESTEP``; ``discrete='exact'`` the
        ``exact=True`` branch (:45-50, matrix exponential of the block matrix; ``scg_discretize_exact`` on the
        matrix cores).  Per-env inertial properties are used when ``randomized_inertial_prop`` is on."""
        self._ensure_engine()
        xs, us, inertial, ldx = self._model_planes(state, input)
        if discrete == 'exact':
            from safe_control_gym_amd.controllers.mpc.mpc_utils import discretize_exact
            A, B = self._engine.linearize(xs, us, inertial, self.num_envs, ldx, 0.0)
            Ad, Bd, _ = discretize_exact(A, B, self.CTRL_TIMESTEP)
            return Ad, Bd
        return self._engine.linearize(xs, us, inertial, self.num_envs, ldx,
                                      self.CTRL_TIMESTEP if discrete else 0.0)

    def _linear_planes(self, x_eval, u_eval, state, input):
        xs, us, inertial, ld = self._model_planes(state, input)
        if ld != self._ld:                                 # bring the linearisation point to the env's own plane stride
            xs, us = self._planes(xs[:, :self.num_envs].T, self.state_dim)[0], self._planes(us[:, :self.num_envs].T, self.action_dim)[0]
            inertial = self._bufs.get('inertial')
        xe = None if x_eval is None else self._planes(x_eval, self.state_dim)[0]
        ue = self._planes(u_eval, self.action_dim, default=np.asarray(self.U_EQ, dtype=np.float64))[0]
        for t in (xe, ue):
            if t is not None and t.shape[1] != self._ld:
                raise ValueError('[ERROR] evaluation points must be (num_envs, dim) rows or planes with the env\'s plane stride.')
        return xe, ue, xs, us, inertial

    def model_linear_dynamics(self, x_eval, u_eval=None, state=None, input=None):
        """Batched ``symbolic.fc_linear_func`` (symbolic_systems.py:93-101): the model linearised about each env's
        (``state``, ``input``) and evaluated at (``x_eval``, ``u_eval``):
        ``f(x, u) + dfdx (x_eval - x) + dfdu (u_eval - u)`` -> [N, nx].  Defaults as in ``linearize``."""
        self._ensure_engine()
        xe, ue, xs, us, inertial = self._linear_planes(x_eval, u_eval, state, input)
        return self._engine.fc_linear(xe, ue, xs, us, inertial, self.num_envs, self._ld)[:, :self.num_envs].T

    def model_linear_step(self, x_eval, u_eval=None, state=None, input=None):
        """Batched ``symbolic.fd_linear_func`` (symbolic_systems.py:102-107): the linearised model integrated over one
        control step from ``x_eval`` under constant ``u_eval`` -> [N, nx].  The ODE is affine, ``z' = A z + c`` with
        ``A = dfdx(x, u)`` and ``c = fc_linear(0, u_eval; x, u)``, so its flow is exact: ``z(dt) = Ad z(0) + Gamma c`` from
        the matrix exponential of ``[[A, c], [0, 0]] dt`` (the reference integrates the same ODE with CVODES)."""
        from safe_control_gym_amd.controllers.mpc.mpc_utils import discretize_exact
        self._ensure_engine()
        n = self.num_envs
        xe, ue, xs, us, inertial = self._linear_planes(x_eval, u_eval, state, input)
        A, _ = self._engine.linearize(xs, us, inertial, n, self._ld, 0.0)
        c = self._engine.fc_linear(None, ue, xs, us, inertial, n, self._ld)[:, :n].T.contiguous()
        z0 = xe[:, :n].T.contiguous()
        return discretize_exact(A, c.reshape(n, self.state_dim, 1), self.CTRL_TIMESTEP, z0=z0, want_matrices=False)[2]

    def quadratic_loss(self, state=None, input=None, x_ref=None, u_ref=None, Q=None, R=None, gradients=True):
        """Batched ``symbolic.loss`` (quadrotor.py:678-682, symbolic_systems.py:94-123) on the device:
        ``l [N]``, ``l_x [N, nx]``, ``l_u [N, nu]`` (+ the constant Hessians) for each env's (state, input).
        ``x_ref`` / ``u_ref``: a vector for all envs (default ``X_GOAL`` row 0 / ``U_GOAL``) or [N, .] rows."""
        self._ensure_engine()
        n, nx, nu = self.num_envs, self.state_dim, self.action_dim
        xs, ldx = self._planes(state, nx, default=self._bufs['state'])
        us, ldu = self._planes(input, nu, default=np.asarray(self.U_EQ, dtype=np.float64))
        if ldx != self._ld:
            xs = self._planes(xs[:, :n].T, nx)[0]
        if ldu != self._ld:
            us = self._planes(us[:, :n].T, nu)[0]
        def ref(v, dim, default):
            v = default if v is None else v
            v = v if isinstance(v, torch.Tensor) else np.asarray(v, dtype=np.float32)
            if v.ndim == 1:
                return torch.as_tensor(v, dtype=torch.float32, device=self.device).contiguous()
            return self._planes(v, dim)[0]
        xg = np.asarray(self.X_GOAL, dtype=np.float64)
        xr = ref(x_ref, nx, xg if xg.ndim == 1 else xg[0])
        ur = ref(u_ref, nu, np.asarray(self.U_GOAL, dtype=np.float64))
        Q = np.asarray(self.Q if Q is None else Q, dtype=np.float64).reshape(nx, nx)
        R = np.asarray(self.R if R is None else R, dtype=np.float64).reshape(nu, nu)
        l, l_x, l_u = self._engine.quadratic_loss(xs, us, xr, ur, Q, R, n, self._ld, want_grad=gradients)
        out = {'l': l[:n]}
        if gradients:
            f32 = dict(dtype=torch.float32, device=self.device)
            out.update({'l_x': l_x[:, :n].T, 'l_u': l_u[:, :n].T,
                        'l_xx': torch.as_tensor(0.5 * (Q + Q.T), **f32).expand(n, nx, nx),
                        'l_uu': torch.as_tensor(0.5 * (R + R.T), **f32).expand(n, nu, nu),
                        'l_xu': torch.zeros((1, 1, 1), **f32).expand(n, nx, nu)})
        return out

    def rollout(self, actions, steps=None, fused=False, out=None):
        """Open-loop rollout: ``steps`` fused step launches issued from one C-ABI call (``scg_rollout``), or with
        ``fused=True`` ONE launch that keeps every env in registers for all the steps (``scg_rollout_fused``; same
        final buffers, ~8x less HBM traffic; lean configurations only, others fall back).

        ``actions``: contiguous float32 device tensor [T, nu, ld] (``ld = env.plane_stride``), used
        cyclically.  Outputs (``env.state``, reward, done, ... buffers) hold the last step's values;
        episode statistics accumulate.  Passive disturbances are drawn in the kernel at every step; an adversary
        needs ``step()`` (its action changes per step).

        ``out``: a ``RolloutTrajectory`` (``env.trajectory_buffer(steps)``) -- step k then writes its observation / reward / done
        (and whatever else the buffer holds) into row k of the buffer instead (``scg_rollout_trajectory``): the rollout yields
        a ``[T, N, ...]`` dataset at the same bytes per env-step (the state ping-pongs through the buffer's observation slabs;
        with ``fused=True`` the envs stay in registers and only the outputs stream out).  The sunk outputs are not written to
        the env's own per-step buffers."""
        self._check_initial_reset()
        self._refuse_while_pending('rollout()')
        if self.adversary_disturbance is not None:
            raise RuntimeError('[ERROR] rollout() cannot take per-step adversary actions; use step().')
        if not (isinstance(actions, torch.Tensor) and actions.dtype == torch.float32 and actions.is_contiguous()
                and actions.ndim == 3 and actions.shape[1:] == (self.action_dim, self._ld) and actions.device == self._bufs['state'].device):
            raise ValueError('[ERROR] rollout() needs a contiguous float32 [T, %d, %d] device tensor.'
                             % (self.action_dim, self._ld))
        steps = actions.shape[0] if steps is None else int(steps)
        sink = None
        if out is not None:
            if out.env_signature != (self.num_envs, self._ld, self.obs_dim) or out.steps < steps:
                raise ValueError('[ERROR] rollout(out=...): the trajectory buffer was made for another env or fewer steps.')
            sink = out.sink_tensors()
        self._own_state()
        io = dict(self._bufs)
        io['reset_state'] = self._reset_state_buf
        self._engine.rollout(io, self.num_envs, self._ld, self._launch_serial(), steps, actions, fused=fused, sink=sink)
        self._bound = False
        self._serial += steps
        k = (steps - 1) % actions.shape[0]
        self._raw_action, self._raw_action_rows = actions[k], False
        if out is not None:
            out.filled = steps
        return out

    def trajectory_buffer(self, steps, constraints=None, info=False):
        """A device buffer for ``rollout(..., out=)``: time-major planes ``[T][obs_dim][ld]`` / ``[T][ld]`` whose ``obs`` /
        ``reward`` / ``done`` (/ ``constraint_values`` ...) attributes are zero-copy ``[T, N, ...]`` views -- the layout of the
        reference's rollout buffer (controllers/ppo/ppo_utils.py:308-336).  ``constraints`` (default: whether the env has any)
        adds the constraint values and the violation flag; ``info=True`` the per-step ``mse`` and ``TimeLimit.truncated``
        (per-step launches only)."""
        self._ensure_engine()
        return RolloutTrajectory(self, int(steps), self.constraints is not None if constraints is None else bool(constraints),
                                 bool(info))

    @property
    def plane_stride(self):
        self._ensure_engine()
        return self._ld

    def set_init_states(self, init_states):
        """Per-env nominal reset states [N, nx] (replaces the single ``init_state`` for every later reset)."""
        t = torch.as_tensor(init_states, dtype=torch.float32, device=self.device)
        if t.shape != (self.num_envs, self.state_dim):
            raise ValueError('[ERROR] init_states must have shape (num_envs, state_dim).')
        self._ensure_engine()
        buf = torch.zeros((self.state_dim, self._ld), dtype=torch.float32, device=self.device)
        buf[:, :self.num_envs] = t.T
        self._reset_state_buf = buf
        self._bound = False

    # ------------------------------------------------------------------ reset / step
    def extend_obs(self, obs, next_step):
        """benchmark_env.py:406-420 for a batch: append the next ``obs_goal_horizon`` reference states (tracking; rows
        ``min(next_step + i, T - 1)``) or the goal (stabilisation) to ``obs`` [N, nx].  ``next_step``: an int or an [N] tensor.
        The step kernel does this itself; the method is for callers that build observations of their own."""
        if not (self.COST == Cost.RL_REWARD and self.obs_goal_horizon > 0):
            return obs
        obs = torch.as_tensor(obs, dtype=torch.float32, device=self.device)
        goal = torch.as_tensor(np.asarray(self.X_GOAL, dtype=np.float32), device=obs.device)
        if self.TASK == Task.TRAJ_TRACKING:
            step = torch.as_tensor(next_step, device=obs.device).reshape(-1, 1).expand(obs.shape[0], 1).long()
            idx = (step + torch.arange(self.obs_goal_horizon, device=obs.device)[None, :]).clamp(max=goal.shape[0] - 1)
            return torch.cat([obs, goal[idx].reshape(obs.shape[0], -1)], dim=1)
        return torch.cat([obs, goal.reshape(1, -1).expand(obs.shape[0], -1)], dim=1)

    def _obs_view(self):
        src = self._bufs['obs'] if 'obs' in self._bufs else self._bufs['state']
        return src[:, :self.num_envs].T

    # ------------------------------------------------------------------ recycled output sets (output_pool)
    @staticmethod
    def _new_output_set(n, ld, od, nx, device):
        """One output set in ONE allocation (utils/recycle.py): obs planes [obs_dim, ld] | reward [n] | done [n] (256-B aligned), the
        state rows of the planes and the views step() returns.  (A function of plain numbers: the pool holds no reference to the env.)"""
        a = od * ld * 4
        r = (4 * n + 255) // 256 * 256
        arena = torch.empty(a + r + n, dtype=torch.uint8, device=device)
        planes = arena[:a].view(torch.float32).view(od, ld)
        rew = arena[a:a + 4 * n].view(torch.float32)
        done = arena[a + r:a + r + n].view(torch.bool)
        state_rows = planes if od == nx else planes[:nx]
        obs = planes.T if ld == n else planes[:, :n].T
        return (planes, rew, done, state_rows, (obs, rew, done, planes)), (planes, rew, done, state_rows, obs), arena.untyped_storage()

    def _own_state(self):
        """Fresh-output mode (``copy_outputs=True``): whoever is about to write the engine's buffers IN PLACE (reset, rollout,
        load_state_dict) first gives the engine private copies of everything the last ``step()`` / ``reset()`` handed out --
        the state buffer in ping-pong mode (it IS the last observation), a kernel-written observation buffer (observation
        disturbance), and the reward / done tensors -- so that a tensor, once returned, is never written again."""
        if not (self._fresh_outputs or self._state_external) or not self._handed_out:
            return                      # nothing of the engine's buffers is in a caller's hands (e.g. rollout() after rollout())
        self._handed_out = False
        b = self._bufs
        if self._state_external and not self._fresh_outputs:
            # views mode after step(out={'obs': ...}): the state (or the observation buffer) lives in the caller's tensor
            self._state_external = False
            if 'obs' in b:
                b['obs'] = b['obs'].clone()
                if 'observation' not in self.disturbances:
                    b['state'] = b['obs'][:self.state_dim]
            else:
                b['state'] = b['state'].clone()
            self._out_views = None
            self._bound = False
            return
        if self._pingpong:
            if 'obs' in b:
                b['obs'] = b['obs'].clone()
                b['state'] = b['obs'][:self.state_dim]
            else:
                b['state'] = b['state'].clone()
        elif 'obs' in b:
            b['obs'] = b['obs'].clone()
        b['reward'], b['done'] = b['reward'].clone(), b['done'].clone()
        self._bound = False

    def reset(self, mask=None):
        """(Re-)initialise all envs (or those where ``mask`` is set).  Returns obs or (obs, info).

        ``reseed_on_reset=True`` (benchmark_env.py:356-362, ``before_reset``: ``self.seed(self.RND_SEED)`` ahead of every reset) makes
        every reset draw the SAME initial states, inertial properties and disturbance streams: the Philox stream is rewound,
        the handle (same key) is kept."""
        self._refuse_while_pending('reset()')
        if self._capture_base is not None:
            raise RuntimeError('[ERROR] reset() inside capture_loop(): record steps only (the envs auto-reset inside the step kernel).')
        if self.RESEED_ON_RESET:
            self.seed(self.RND_SEED)
        self._ensure_engine()
        self.initial_reset = True
        self._raw_action, self._raw_action_rows = None, False
        for mode in self.disturbances:
            self.disturbances[mode].reset(self)
        if self.adversary_disturbance is not None:
            self._adv_raw = None
            self._adv_fresh = False
            if self._bufs is not None:                     # an env that was reset starts without the adversary's last input
                for key in ('action_disturbance', 'dynamics_force'):
                    if key in self._bufs:
                        self._bufs[key].zero_()
                self._adv_force_live = False
        m = None
        if mask is not None:
            m = torch.zeros(self._ld, dtype=torch.uint8, device=self.device)
            m[:self.num_envs] = torch.as_tensor(mask, device=self.device).to(torch.uint8)
        self._own_state()
        rb = self._bufs
        if self.COMPENSATED_STATE == 'attitude':
            # scg_reset zeroes nx residual planes; this mode has 6: zero them here (masked lanes only)
            rb = {k: v for k, v in self._bufs.items() if k != 'state_lo'}
            if m is None:
                self._bufs['state_lo'].zero_()
            else:
                self._bufs['state_lo'][:, m.bool()] = 0
        self._engine.reset(rb, m, self._reset_state_buf, self._serial, self.num_envs, self._ld)
        self._bound = False
        self._serial += 1
        self._handed_out = True
        obs = self._obs_view()
        if self.COPY_OUTPUTS and not self._pingpong:          # ping-pong: nobody writes this buffer again (_own_state)
            obs = obs.clone()
        if self.INFO_IN_RESET:
            return obs, self._get_reset_info()
        return obs

    def _get_reset_info(self):
        info = {'x_reference': self.X_GOAL, 'u_reference': self.U_GOAL, 'symbolic_model': self.symbolic,
                'ctrl_timestep': self.CTRL_TIMESTEP, 'ctrl_freq': self.CTRL_FREQ,
                'episode_len_sec': self.EPISODE_LEN_SEC}
        if self.constraints is not None:
            info['symbolic_constraints'] = self.constraints.get_all_symbolic_models()
            info['constraint_values'] = self.constraints.get_values(self, only_state=True)   # benchmark_env.py:381-382
        else:
            info['symbolic_constraints'] = {}
        info['initial_state_randomization'] = self.INIT_STATE_RAND_INFO if self.RANDOMIZED_INIT else {}
        info['inertial_prop_randomization'] = self.INERTIAL_PROP_RAND_INFO if self.RANDOMIZED_INERTIAL_PROP else {}
        info['disturbances'] = self.DISTURBANCES
        return info

    def _load_action(self, action):
        """-> (tensor, rows): the raw action as the kernel will read it.  A float32 device tensor [N, nu] (what a policy
        emits) is read in place, one 4 nu-byte vector per lane (``rows=True``); a tensor that already IS SoA planes
        (``env.action_buffer`` / its transpose, or [nu, ld]) is read in place as planes; anything else (host arrays, other
        dtypes, odd strides) is copied into the env's action planes."""
        buf = self._bufs['action']
        n, nu = self.num_envs, self.action_dim
        if isinstance(action, torch.Tensor):
            if action.data_ptr() == buf.data_ptr():
                return buf, False                      # env.action_buffer (or its [N, nu] view): the planes themselves
            if (action.shape == (nu, self._ld) and action.dtype == torch.float32 and action.is_contiguous()
                    and action.device == buf.device):
                return action, False                   # already SoA with the engine's plane stride: zero copy
            if (action.shape == (n, nu) and action.dtype == torch.float32 and action.is_contiguous()
                    and action.device == buf.device and action.data_ptr() % (4 * nu) == 0):
                return action, True                    # row-major rows, read as vectors: zero copy
            a = action.to(device=self.device, dtype=torch.float32)
        else:
            a = torch.as_tensor(np.asarray(action, dtype=np.float32), device=self.device)
        if a.ndim == 1:
            a = a.reshape(1, -1) if n == 1 else a.reshape(n, -1)
        if a.shape == (n, nu):
            buf[:, :n].copy_(a.T)
        elif a.shape == (nu, n):
            buf[:, :n].copy_(a)
        else:
            raise ValueError('[ERROR] action must have shape (num_envs, action_dim) = (%d, %d), got %s'
                             % (n, nu, tuple(a.shape)))
        return buf, False

    def step(self, action, out=None):
        """Advance every env by one control step (quadrotor.py:507-558 + benchmark_env.py:422-463).

        ``out`` (optional): ``{'obs': planes, 'reward': t, 'done': t}`` (any subset) -- tensors of the CALLER's that this step's
        outputs are written into by the kernel itself, e.g. row t of a device rollout buffer (``PPOBuffer(env=...)``:
        controllers/ppo/ppo_utils.py:308-336 without the per-step copy of ``push``): ``obs`` a contiguous float32
        ``[obs_dim, plane_stride]`` plane buffer (it becomes the env's state buffer: the state ping-pongs into it),
        ``reward`` float32 ``[>= N]``, ``done`` bool / uint8 ``[>= N]``.  The returned ``obs`` / ``rew`` / ``done`` are views of them.

        OWNERSHIP OF THE OUTPUTS.  With ``copy_outputs=True`` (the default of ``info_level='full'``, i.e. of a plain
        ``make()``) ``obs``, ``rew`` and ``done`` are fresh tensors, as the reference's are: the usual RL loop keeps the
        previous ``obs`` until after the next ``step()`` (controllers/ppo/ppo.py:259-276).  With ``copy_outputs=False``
        (default of ``info_level='minimal'``) they are zero-copy VIEWS of the engine's planes -- valid until the next
        ``step()`` / ``rollout()`` / ``reset()``, ``.clone()`` what must outlive it.  Fresh outputs cost no copy: the state
        ping-pongs -- the kernel writes the new state (``scg_step_io.state_out``), reward and done straight into newly allocated
        tensors, the state buffer IS the observation handed out and the next step's input, and nothing writes it again
        (``_own_state``; writing ``env.state[...]`` yourself writes into the last observation).  Only an observation
        disturbance needs an obs buffer of its own.  The tensors inside ``info`` are always views of per-step buffers."""
        self._refuse_while_pending('step()')
        return self._step_outputs(self._step_launch(action, False, out))

    def step_async(self, action, out=None):
        """The first half of ``VecEnv.step`` (vec_env.py:47-59; ``SubprocVecEnv.step_async`` sends the actions to its workers
        and returns, subproc_vec_env.py:50-55): everything queued on the current stream so far -- the policy that produced
        ``action`` -- happens before the step; the step kernel itself goes to THIS ENV'S OWN launch queue
        (``scg_step_async``) and the call returns.  ``step_wait()`` makes the current stream wait for it (on the device; the
        host never blocks) and hands out the outputs.  An env that is one shard of a batch (``make_vec_envs(...,
        num_pipelines=2)``) thereby overlaps its kernel's ramp-up / drain, and its policy, with the other shard's kernel.
        Call both from the same current stream; ``pipeline_stream`` is the queue itself, for callers that run a shard's policy
        on it (then neither call adds an event)."""
        self._refuse_while_pending('step_async()')
        self._async_outs = self._step_launch(action, True, out)

    def step_wait(self):
        """The second half (vec_env.py:61-72): ``(obs, rew, done, info)`` of the pending ``step_async()``; work queued on the
        current stream from here on sees them complete."""
        if self._async_outs is None:
            raise RuntimeError('[ERROR] step_wait() without a pending step_async().')
        outs, self._async_outs = self._async_outs, None
        self._engine.wait()
        # what the kernel was reading (the action, the previous state buffer) may be released now: allocations on the
        # current stream are ordered behind the kernel from here on
        self._inflight = None
        return self._step_outputs(outs)

    @property
    def pipeline_stream(self):
        """This env's own launch queue as a ``torch.cuda.Stream`` (``scg_step_queue``)."""
        self._ensure_engine()
        return self._engine.queue_stream()

    def _check_out(self, out):
        """Validated (obs planes | None, reward | None, done | None) of step(out=...)."""
        n, f32 = self.num_envs, torch.float32
        obs_t, rew_t, done_t = out.get('obs'), out.get('reward'), out.get('done')
        if set(out) - {'obs', 'reward', 'done'}:
            raise ValueError("[ERROR] step(out=...): keys are 'obs', 'reward', 'done'.")
        if obs_t is not None and not (isinstance(obs_t, torch.Tensor) and obs_t.dtype == f32 and obs_t.is_contiguous()
                                      and tuple(obs_t.shape) == (self.obs_dim, self._ld) and obs_t.device == self._bufs['state'].device):
            raise ValueError('[ERROR] step(out=...): obs must be a contiguous float32 (%d, %d) device tensor (obs_dim planes of '
                             'env.plane_stride).' % (self.obs_dim, self._ld))
        if rew_t is not None and not (isinstance(rew_t, torch.Tensor) and rew_t.dtype == f32 and rew_t.ndim == 1 and rew_t.is_contiguous()
                                      and rew_t.shape[0] >= n and rew_t.device == self._bufs['state'].device):
            raise ValueError('[ERROR] step(out=...): reward must be a contiguous float32 (>= %d,) device tensor.' % n)
        if done_t is not None and not (isinstance(done_t, torch.Tensor) and done_t.dtype in (torch.bool, torch.uint8) and done_t.ndim == 1
                                       and done_t.is_contiguous() and done_t.shape[0] >= n and done_t.device == self._bufs['state'].device):
            raise ValueError('[ERROR] step(out=...): done must be a contiguous bool / uint8 (>= %d,) device tensor.' % n)
        if obs_t is not None and (self.COMPENSATED_STATE or self.adversary_disturbance is not None):
            raise ValueError('[ERROR] step(out=...): an obs target is not available in the precise modes or with an adversary.')
        return obs_t, rew_t, done_t

    def _step_launch(self, action, asynchronous, out=None):
        self._check_initial_reset()
        act, rows = self._load_action(action)
        self._raw_action, self._raw_action_rows = act, rows
        b = self._bufs
        outs = None
        new_state = state_rows = None
        if asynchronous:
            # the kernel runs on another queue than the one the caching allocator knows these tensors by: keep what it READS
            # alive until step_wait() has ordered the current stream behind it
            self._inflight = (act, b['state'], b.get('obs'))
        if out is not None:
            # the caller's tensors receive this step's outputs (a row of a rollout buffer): the fresh-output mechanism with the
            # allocation done by the caller
            obs_t, rew_t, done_t = self._check_out(out)
            if rew_t is None:
                rew_t = torch.empty(self.num_envs, dtype=torch.float32, device=self.device) if self._fresh_outputs else b['reward']
            if done_t is None:
                done_t = torch.empty(self.num_envs, dtype=torch.bool, device=self.device) if self._fresh_outputs else b['done']
            if obs_t is None and self._fresh_outputs:
                obs_t = torch.empty((self.obs_dim, self._ld), dtype=torch.float32, device=self.device)
            if obs_t is None:
                outs = (b.get('obs'), rew_t, done_t)           # state / obs stay where they are (views)
            elif 'observation' in self.disturbances:
                outs = (obs_t, rew_t, done_t)                  # a kernel-written observation buffer of its own; the state stays in place
                b['obs'] = obs_t
            else:
                new_state = obs_t                              # the state ping-pongs into the caller's planes
                outs = (obs_t if 'obs' in b else None, rew_t, done_t)
            if self._fresh_outputs:
                b['reward'], b['done'] = rew_t, done_t
            else:
                # views mode: the env keeps its own reward / done planes for later plain steps (the io is re-bound), and a state /
                # observation that now lives in the caller's tensor is copied home before anything writes it in place
                self._bound_dirty = True
                self._state_external = self._state_external or obs_t is not None
            self._out_views = None                             # the cached views of the lean tier point at the old buffers
        elif self._state_external:
            self._own_state()                                  # a plain (in-place) step after step(out=...): private planes first
            b = self._bufs
        elif self._fresh_outputs:
            # copy_outputs: this step's outputs get tensors of their own, written by the kernel (no clone afterwards)
            f32 = torch.float32
            slot_views = slot = None
            if self._ring is not None:
                fresh, rew_t, done_t, state_rows, slot_views = self._ring[self._ring_pos]
                self._ring_pos = (self._ring_pos + 1) % len(self._ring)
            elif self._pool is not None and (slot := self._pool.take()) is not None:
                fresh, rew_t, done_t, state_rows, slot_views = slot
                slot = None
            else:
                fresh = torch.empty((self.obs_dim, self._ld), dtype=f32, device=self.device)
                rew_t = torch.empty(self.num_envs, dtype=f32, device=self.device)       # exactly n: the kernel writes lanes < n only
                done_t = torch.empty(self.num_envs, dtype=torch.bool, device=self.device)
            if self._pingpong:
                # the new state is written into `fresh` (with the goal rows behind it, if any); the buffer read this step
                # stays untouched: it is the observation the previous call handed out
                new_state = fresh
                outs = (fresh if 'obs' in b else None, rew_t, done_t, slot_views)
            else:
                outs = (fresh, rew_t, done_t, slot_views)
                b['obs'] = fresh
            b['reward'], b['done'] = rew_t, done_t
        if self.adversary_disturbance is not None:
            # adversary inputs (benchmark_env.py:256-266, quadrotor.py:529-536,837-838) arrive as planes; the passive
            # disturbances are generated inside the kernel (scg_set_disturbances)
            self._adversary_planes()
        if not self._bound:
            io = dict(b)
            io['reset_state'] = self._reset_state_buf
            self._engine.bind_static(io, self._ld)
            self._bound = True
        self._engine.step_bound(act, self.num_envs, self._launch_serial(), action_rows=rows, outs=outs, state_out=new_state,
                                asynchronous=asynchronous)
        self._serial += 1
        self._handed_out = True
        if new_state is not None:
            if 'obs' in b:
                b['obs'] = new_state
            if state_rows is not None:
                b['state'] = state_rows                          # output pool / ring: the set's own view, made once
            else:
                b['state'] = new_state if new_state.shape[0] == self.state_dim else new_state[:self.state_dim]
        if self._bound_dirty:
            # views mode after step(out=...): later plain steps write the env's own reward / done planes again
            self._bound_dirty = False
            if self._bound:
                self._engine.restore_outputs(b['reward'], b['done'])
        return new_state, outs

    def _step_outputs(self, launched):
        new_state, outs = launched
        b = self._bufs
        if outs is not None and len(outs) > 3 and outs[3] is not None:
            obs, rew, done = outs[3][0], outs[3][1], outs[3][2]   # output pool / ring: the set's own views, made once
        elif outs is not None:
            src = new_state if new_state is not None else outs[0]
            if src is None:
                src = b['obs'] if 'obs' in b else b['state']
            rew, done = outs[1], outs[2]
            n = self.num_envs
            if self._fresh_outputs:
                # new tensors every step: nothing to remember (and nothing to look up); a plane stride equal to n needs no slice
                obs = src.T if src.shape[1] == n else src[:, :n].T
                if rew.shape[0] != n:
                    rew = rew[:n]
                if done.shape[0] != n:
                    done = done[:n]
            else:
                # the views handed out, remembered per (planes, reward row, done row): a rollout buffer presents the same rows every
                # T steps, and four tensor views cost ~8 us of host time per step
                key = (src.data_ptr(), rew.data_ptr(), done.data_ptr())
                views = self._out_view_cache.get(key)
                if views is None or views[3] is not src:
                    if len(self._out_view_cache) >= 8192:
                        self._out_view_cache.clear()
                    views = (src.T if src.shape[1] == n else src[:, :n].T, rew if rew.shape[0] == n else rew[:n],
                             done if done.shape[0] == n else done[:n], src)
                    self._out_view_cache[key] = views
                obs, rew, done = views[0], views[1], views[2]
        else:
            if self._out_views is None:
                self._out_views = ((b['obs'] if 'obs' in b else b['state'])[:, :self.num_envs].T, b['reward'][:self.num_envs],
                                   b['done'][:self.num_envs])
            obs, rew, done = self._out_views
            if self.COPY_OUTPUTS:
                obs, rew, done = obs.clone(), rew.clone(), done.clone()
        info = dict(self._info_views)
        if info.get('constraint_values', 0) is None:                 # folded constraints (|y| - bound): needs a max()
            info['constraint_values'] = self.constraints.fold(b['constraint_values'][:, :self.num_envs].T)
        if self.INFO_LEVEL == 'full':
            # (not `info` itself: info -> info['n'] -> info would be a reference cycle, and `done` would live until the next
            # garbage collection instead of until the caller drops it)
            info['n'] = LazyInfoList(dict(info), done)
        return obs, rew, done, info

    @property
    def adv_action(self):
        """The adversary's pending input as the reference holds it: clip(action, -1, 1) * scale + offset (benchmark_env.py:262-264)."""
        if self._adv_raw is None:
            return None
        return self._adv_raw.clamp(-1.0, 1.0) * self.adversary_disturbance_scale + self.adversary_disturbance_offset

    def _adv_fill(self, dst, src):
        """dst [k, n] (a view of planes) = clip(src [n, k], -1, 1).T * scale + offset: scale, then clip to +-|scale| (the same
        numbers), in place."""
        scale, offset = float(self.adversary_disturbance_scale), float(self.adversary_disturbance_offset)
        torch.mul(src.T, scale, out=dst)
        dst.clamp_(-abs(scale), abs(scale))
        if offset != 0.0:
            dst.add_(offset)

    def _adversary_planes(self):
        """The adversary's pending action into its static planes, on the current stream, right before the launch that reads them."""
        n, b = self.num_envs, self._bufs
        if self.adversary_disturbance == 'action':
            if self._adv_fresh:                            # (the action adversary's last input stays in force until the next one)
                self._adv_fill(b['action_disturbance'][:, :n], self._adv_raw)
                self._adv_fresh = False
        elif self._adv_raw is not None:
            # world-frame force [3][ld] from the mode-dimensional adversary action (quadrotor.py:529-548)
            f, out = self._adv_raw, b['dynamics_force']
            self._adv_raw = None                           # cleared after use (:534-536)
            dim = self.DISTURBANCE_MODES['dynamics']['dim']
            if dim == 1:
                self._adv_fill(out[(0 if self.NAME == 'cartpole' else 2):(1 if self.NAME == 'cartpole' else 3), :n], f)
            elif dim == 2:
                self._adv_fill(out[0:3:2, :n], f)
            else:
                self._adv_fill(out[:, :n], f)
            self._adv_force_live = True
        elif self._adv_force_live:                         # no new input: the force of the previous step must not act again
            b['dynamics_force'].zero_()
            self._adv_force_live = False

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    def render(self, mode='human'):
        raise NotImplementedError('[ERROR] the batched engine does not render.')

    # ------------------------------------------------------------------ reference trajectories
    def _generate_trajectory(self, traj_type='figure8', traj_length=10.0, num_cycles=1, traj_plane='xy',
                             position_offset=np.array([0, 0]), scaling=1.0, sample_time=0.01):
        """Sampled planar reference (benchmark_env.py:465-519), closed forms evaluated for all samples at once.

        Returns (pos (T,3), vel (T,3), speed (T,1)) sampled at ``np.arange(0, traj_length, sample_time)``.
        """
        if traj_type not in ('circle', 'square', 'figure8'):
            raise ValueError('Trajectory type should be one of [circle, square, figure8].')
        axes = ['x', 'y', 'z']
        if traj_plane[0] in axes and traj_plane[1] in axes and traj_plane[0] != traj_plane[1]:
            ia, ib = axes.index(traj_plane[0]), axes.index(traj_plane[1])
        else:
            raise ValueError('Trajectory plane should be in form of ab, where a and b can be {x, y, z}.')
        period = traj_length / num_cycles
        t = np.arange(0, traj_length, sample_time)
        w = 2.0 * np.pi / period
        if traj_type == 'circle':                    # benchmark_env.py:613-617
            a, b = scaling * np.cos(w * t), scaling * np.sin(w * t)
            ad, bd = -scaling * w * np.sin(w * t), scaling * w * np.cos(w * t)
        elif traj_type == 'figure8':                 # :587-591
            a, b = scaling * np.sin(w * t), scaling * np.sin(w * t) * np.cos(w * t)
            ad, bd = scaling * w * np.cos(w * t), scaling * w * (np.cos(w * t) ** 2 - np.sin(w * t) ** 2)
        else:                                        # :640-673
            seg = period / 4.0
            speed = scaling / seg
            cyc = t % period
            seg_t = cyc % seg
            idx = np.floor(cyc / seg).astype(int)
            s = speed * seg_t
            zero, full = np.zeros_like(t), np.full_like(t, scaling)
            a = np.select([idx == 0, idx == 1, idx == 2, idx == 3], [zero, -s, -full, -scaling + s])
            b = np.select([idx == 0, idx == 1, idx == 2, idx == 3], [s, full, scaling - s, zero])
            ad = np.select([idx == 0, idx == 1, idx == 2, idx == 3], [zero, zero - speed, zero, zero + speed])
            bd = np.select([idx == 0, idx == 1, idx == 2, idx == 3], [zero + speed, zero, zero - speed, zero])
        pos = np.zeros((len(t), 3))
        vel = np.zeros((len(t), 3))
        pos[:, ia], vel[:, ia] = a + position_offset[0], ad
        pos[:, ib], vel[:, ib] = b + position_offset[1], bd
        return pos, vel, np.linalg.norm(vel, axis=1, keepdims=True)
