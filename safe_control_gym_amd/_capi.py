"""ctypes binding of ``include/scg_rollout.h`` (the C ABI of the HIP rollout engine).

The product path has NO fallback: if ``lib/libscg_rollout.so`` is missing or a call fails, this
module raises.  Struct layouts mirror the header field for field; ``load()`` cross-checks
``ctypes.sizeof`` against the ``scg_sizeof_*`` exports of the library it opened.
"""
import ctypes as C
import os

SCG_ABI_VERSION = 6
SCG_MAX_NX = 12
SCG_MAX_NU = 4
SCG_MAX_NV = 16
SCG_MAX_CONSTRAINT_ROWS = 64
SCG_MAX_QUADRATIC = 4
SCG_MAX_GOAL_HORIZON = 8
SCG_MAX_DISTURBANCES = 4

SCG_QUAD_1D, SCG_QUAD_2D, SCG_QUAD_3D, SCG_CARTPOLE = 1, 2, 3, 4
SCG_RK4, SCG_EULER = 0, 1
SCG_STABILIZATION, SCG_TRAJ_TRACKING = 0, 1
SCG_RL_REWARD, SCG_QUADRATIC = 0, 1
SCG_DIST_ACTION, SCG_DIST_DYNAMICS, SCG_DIST_OBSERVATION = 0, 1, 2
SCG_DIST_IMPULSE, SCG_DIST_STEP, SCG_DIST_UNIFORM, SCG_DIST_WHITE_NOISE, SCG_DIST_PERIODIC = 0, 1, 2, 3, 4
(SCG_RAND_UNIFORM, SCG_RAND_NORMAL, SCG_RAND_EXPONENTIAL, SCG_RAND_LAPLACE, SCG_RAND_LOGISTIC, SCG_RAND_TRIANGULAR,
 SCG_RAND_RANDINT, SCG_RAND_CHOICE, SCG_RAND_RAYLEIGH, SCG_RAND_GUMBEL, SCG_RAND_WEIBULL, SCG_RAND_PARETO, SCG_RAND_POWER,
 SCG_RAND_LOGNORMAL, SCG_RAND_CAUCHY, SCG_RAND_GAMMA, SCG_RAND_BETA, SCG_RAND_CHOICE_P) = range(18)

_dbl = C.c_double
_i32 = C.c_int32
_fp = C.c_void_p          # device pointers travel as plain addresses


class ScgConfig(C.Structure):
    _fields_ = [
        ('struct_size', C.c_uint32), ('abi_version', C.c_uint32),
        ('system', _i32), ('integrator', _i32), ('substeps', _i32), ('task', _i32), ('cost', _i32),
        ('ctrl_steps', _i32),
        ('dt', _dbl),
        ('mass', _dbl), ('gravity', _dbl), ('arm', _dbl), ('ixx', _dbl), ('iyy', _dbl), ('izz', _dbl),
        ('kf', _dbl), ('km', _dbl),
        ('pwm2rpm_scale', _dbl), ('pwm2rpm_const', _dbl), ('pwm_min', _dbl), ('pwm_max', _dbl),
        ('pole_length', _dbl), ('pole_mass', _dbl), ('cart_mass', _dbl),
        ('normalized_action', _i32), ('motor_map', _i32),
        ('norm_act_scale', _dbl), ('hover_thrust', _dbl), ('action_scale', _dbl),
        ('act_low', _dbl * SCG_MAX_NU), ('act_high', _dbl * SCG_MAX_NU),
        ('u_goal', _dbl * SCG_MAX_NU),
        ('x_goal', _dbl * SCG_MAX_NX),
        ('rew_state_weight', _dbl * SCG_MAX_NX), ('rew_act_weight', _dbl * SCG_MAX_NU),
        ('rew_exponential', _i32), ('reserved0', _i32),
        ('Q', _dbl * (SCG_MAX_NX * SCG_MAX_NX)),
        ('R', _dbl * (SCG_MAX_NU * SCG_MAX_NU)),
        ('mse_weight', _dbl * SCG_MAX_NX),
        ('done_on_out_of_bound', _i32), ('done_on_violation', _i32),
        ('oob_low', _dbl * SCG_MAX_NX), ('oob_high', _dbl * SCG_MAX_NX),
        ('goal_tolerance', _dbl),
        ('use_constraint_penalty', _i32), ('obs_goal_horizon', _i32),
        ('constraint_penalty', _dbl),
        ('randomized_init', _i32), ('auto_reset', _i32),
        ('init_state', _dbl * SCG_MAX_NX),
        ('init_rand_low', _dbl * SCG_MAX_NX), ('init_rand_high', _dbl * SCG_MAX_NX),
        ('randomized_inertial', _i32), ('reserved2', _i32),
        ('inertial_rand_low', _dbl * 4), ('inertial_rand_high', _dbl * 4),
        ('ground_effect', _i32), ('drag', _i32),
        ('gnd_eff_coeff', _dbl), ('prop_radius', _dbl), ('gnd_eff_h_clip', _dbl), ('drag_coeff_xy', _dbl), ('drag_coeff_z', _dbl),
        ('seed', C.c_uint64),
        ('env_index_base', C.c_int64),
        ('init_rand_kind', _i32 * SCG_MAX_NX), ('init_rand_p', (_dbl * 3) * SCG_MAX_NX),
        ('inertial_rand_kind', _i32 * 4), ('inertial_rand_p', (_dbl * 3) * 4),
        ('rollout_queues', _i32), ('ref_in_lds', _i32), ('rollout_split_min', C.c_int64),
        ('grid_cap_per_cu', _i32), ('rollout_graph', _i32),
        ('lone_launch_forms', _i32), ('reserved3', _i32),
    ]


class ScgConstraintRow(C.Structure):
    _fields_ = [
        ('kind', _i32), ('strict', _i32), ('has_tolerance', _i32), ('p_index', _i32),
        ('tolerance', _dbl), ('b', _dbl),
        ('coef', _dbl * SCG_MAX_NV),
    ]


class ScgStepIO(C.Structure):
    _fields_ = [
        ('struct_size', C.c_uint32), ('action_rows', C.c_uint32),
        ('ld', C.c_int64),
        ('action', _fp), ('action_disturbance', _fp), ('dynamics_force', _fp), ('obs_noise', _fp),
        ('reset_state', _fp),
        ('inertial', _fp), ('state', _fp), ('state_lo', _fp), ('state_out', _fp), ('step_count', _fp), ('ep_return', _fp), ('ep_length', _fp),
        ('episode_id', _fp),
        ('obs', _fp), ('reward', _fp), ('done', _fp), ('constraint_values', _fp),
        ('constraint_violation', _fp), ('truncated', _fp), ('goal_reached', _fp), ('mse', _fp),
        ('terminal_record', _fp), ('terminal_stride', C.c_int64), ('stats', _fp),
        ('reset_serial', C.c_uint32), ('state_lo_kind', C.c_uint32),
        ('serial_word', _fp),
    ]


class ScgTrajectorySink(C.Structure):
    """scg_trajectory_sink: step k of a multi-step rollout writes its outputs at base + k * step stride (elements)."""
    _fields_ = [
        ('struct_size', C.c_uint32), ('flags', C.c_uint32),
        ('obs', _fp), ('obs_step_stride', C.c_int64),
        ('reward', _fp), ('reward_step_stride', C.c_int64),
        ('done', _fp), ('done_step_stride', C.c_int64),
        ('constraint_values', _fp), ('constraint_values_step_stride', C.c_int64),
        ('constraint_violation', _fp), ('constraint_violation_step_stride', C.c_int64),
        ('truncated', _fp), ('truncated_step_stride', C.c_int64),
        ('mse', _fp), ('mse_step_stride', C.c_int64),
    ]


SCG_SINK_KEEP_STATE_IN_SINK = 1


class ScgDisturbance(C.Structure):
    _fields_ = [
        ('mode', _i32), ('kind', _i32), ('step_offset', _i32), ('max_step', _i32), ('duration', _i32), ('has_mask', _i32),
        ('magnitude', _dbl), ('decay_rate', _dbl), ('scale', _dbl), ('frequency', _dbl),
        ('low', _dbl * SCG_MAX_NX), ('high', _dbl * SCG_MAX_NX), ('mask', _dbl * SCG_MAX_NX),
    ]


SCG_MAX_COPY_SEGMENTS = 8


class ScgCopySegment(C.Structure):
    """``scg_copy_segment`` (include/scg_rollout.h): one strided rows x cols float32 copy of ``scg_copy_segments``."""
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('rows', C.c_int64), ('cols', C.c_int64),
                ('src_row_stride', C.c_int64), ('src_col_stride', C.c_int64), ('dst_row_stride', C.c_int64), ('dst_col_stride', C.c_int64)]


EXPORTS = (
    'scg_create', 'scg_destroy', 'scg_set_reference', 'scg_set_constraints', 'scg_set_disturbances', 'scg_reset', 'scg_step', 'scg_rollout', 'scg_rollout_fused',
    'scg_step_async', 'scg_step_wait', 'scg_step_queue', 'scg_rollout_trajectory', 'scg_sizeof_trajectory_sink',
    'scg_eval_constraints', 'scg_gae', 'scg_normalize_advantages', 'scg_copy_segments', 'scg_model_eval', 'scg_model_flow', 'scg_linearize', 'scg_fc_linear', 'scg_discretize_exact', 'scg_quadratic_loss', 'scg_mean_std_normalize', 'scg_reward_normalize', 'scg_mean_std_moments', 'scg_mean_std_normalize_sums', 'scg_reward_moments', 'scg_reward_normalize_sums', 'scg_num_constraints', 'scg_obs_dim', 'scg_kernel_info', 'scg_rollout_graph_info', 'scg_last_error',
    'scg_version', 'scg_sizeof_config', 'scg_sizeof_step_io', 'scg_sizeof_constraint_row', 'scg_sizeof_disturbance',
)


class ScgError(RuntimeError):
    """A C-ABI call returned a negative status."""


_LIB = None


def library_path():
    """In-tree library; SCG_ROLLOUT_LIB overrides it (used by tools/tune.py to A/B kernel builds)."""
    override = os.environ.get('SCG_ROLLOUT_LIB')
    if override:
        return override
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libscg_rollout.so')


def load():
    """Open the in-tree HIP library (once).  Raises if it is absent: there is no CPU fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ScgError('HIP extension %s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                       '(or safe_control_gym_amd/csrc/build.sh).  The engine has no CPU fallback.' % path)
    # The library must share ONE HIP runtime with PyTorch (streams and device pointers cross the ABI).
    # torch ships its own libamdhip64.so with the same soname as /opt/rocm's: whichever is loaded first
    # serves both, so make sure it is torch's.
    import torch
    torch_hip = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
    if os.path.exists(torch_hip):
        C.CDLL(torch_hip, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(path)
    lib.scg_last_error.restype = C.c_char_p
    lib.scg_create.argtypes = [C.POINTER(ScgConfig), C.POINTER(C.c_void_p)]
    lib.scg_destroy.argtypes = [C.c_void_p]
    lib.scg_set_reference.argtypes = [C.c_void_p, C.POINTER(_dbl), _i32]
    lib.scg_set_constraints.argtypes = [C.c_void_p, C.POINTER(ScgConstraintRow), _i32, C.POINTER(_dbl), _i32]
    lib.scg_set_disturbances.argtypes = [C.c_void_p, C.POINTER(ScgDisturbance), _i32]
    lib.scg_reset.argtypes = [C.c_void_p, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_uint32, C.c_int64,
                              C.c_int64, C.c_void_p]
    lib.scg_step.argtypes = [C.c_void_p, C.POINTER(ScgStepIO), C.c_int64, C.c_void_p]
    lib.scg_rollout.argtypes = [C.c_void_p, C.POINTER(ScgStepIO), C.c_int64, _i32, C.c_int64, _i32, C.c_void_p]
    lib.scg_rollout_fused.argtypes = [C.c_void_p, C.POINTER(ScgStepIO), C.c_int64, _i32, C.c_int64, _i32, C.c_void_p]
    lib.scg_step_async.argtypes = [C.c_void_p, C.POINTER(ScgStepIO), C.c_int64, C.c_void_p]
    lib.scg_step_wait.argtypes = [C.c_void_p, C.c_void_p]
    lib.scg_step_queue.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.scg_rollout_trajectory.argtypes = [C.c_void_p, C.POINTER(ScgStepIO), C.POINTER(ScgTrajectorySink), C.c_int64, _i32, C.c_int64,
                                           _i32, _i32, C.c_void_p]
    lib.scg_eval_constraints.argtypes = [C.c_void_p, _fp, _fp, _fp, _fp, C.c_int64, C.c_int64, C.c_void_p]
    lib.scg_gae.argtypes = [_fp] * 10 + [_i32, C.c_int64, C.c_int64, C.c_float, C.c_float, _i32, C.c_void_p]
    lib.scg_normalize_advantages.argtypes = [_fp, _fp, _i32, C.c_int64, C.c_int64, C.c_float, C.c_void_p]
    lib.scg_copy_segments.argtypes = [C.POINTER(ScgCopySegment), _i32, C.c_void_p]
    lib.scg_model_eval.argtypes = [C.c_void_p, _fp, _fp, _fp, _fp, _fp, C.c_int64, C.c_int64, C.c_void_p]
    lib.scg_model_flow.argtypes = [C.c_void_p, _fp, _fp, _fp, C.c_double, C.c_double, C.c_double, _i32, _fp, _fp, _fp, _fp,
                                   C.c_int64, C.c_int64, C.c_void_p]
    lib.scg_linearize.argtypes = [C.c_void_p, _fp, _fp, _fp, _fp, _fp, _dbl, C.c_int64, C.c_int64, C.c_void_p]
    lib.scg_fc_linear.argtypes = [C.c_void_p, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int64, C.c_int64, C.c_void_p]
    lib.scg_discretize_exact.argtypes = [_fp, _fp, _i32, _i32, _dbl, _fp, _fp, _fp, _fp, _fp, C.c_int64, C.c_void_p]
    lib.scg_quadratic_loss.argtypes = [C.c_void_p, C.POINTER(_dbl), C.POINTER(_dbl), _fp, _fp, _fp, C.c_int64, _fp, C.c_int64,
                                       _fp, _fp, _fp, C.c_int64, C.c_int64, C.c_void_p]
    lib.scg_mean_std_normalize.argtypes = [_fp, _fp, _i32, C.c_int64, C.c_int64, C.c_int64, _fp, _fp, _fp, C.c_float, _dbl,
                                           C.c_void_p]
    lib.scg_reward_normalize.argtypes = [_fp, _fp, _fp, _fp, C.c_int64, _fp, _fp, _fp, C.c_float, C.c_float, _dbl, _i32,
                                         C.c_void_p]
    lib.scg_mean_std_moments.argtypes = [_fp, _i32, C.c_int64, C.c_int64, _fp, _fp, _fp, C.c_void_p]
    lib.scg_mean_std_normalize_sums.argtypes = [_fp, _fp, _i32, C.c_int64, C.c_int64, C.c_int64, _fp, _fp, _fp, _dbl, C.c_float, _dbl, C.c_void_p]
    lib.scg_reward_moments.argtypes = [_fp, _fp, _fp, C.c_int64, _fp, _fp, _fp, C.c_float, C.c_void_p]
    lib.scg_reward_normalize_sums.argtypes = [_fp, _fp, C.c_int64, _fp, _fp, _fp, _dbl, C.c_float, _dbl, C.c_void_p]
    lib.scg_num_constraints.argtypes = [C.c_void_p]
    lib.scg_obs_dim.argtypes = [C.c_void_p]
    lib.scg_kernel_info.argtypes = [C.c_void_p, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.c_int64]
    lib.scg_rollout_graph_info.argtypes = [C.c_void_p, C.POINTER(_i32), C.POINTER(C.c_int64)]
    for name in EXPORTS:
        if name not in ('scg_last_error',):
            getattr(lib, name).restype = C.c_int
    if lib.scg_version() != SCG_ABI_VERSION:
        raise ScgError('libscg_rollout.so ABI %d != binding ABI %d' % (lib.scg_version(), SCG_ABI_VERSION))
    for struct, fn in ((ScgConfig, lib.scg_sizeof_config), (ScgStepIO, lib.scg_sizeof_step_io),
                       (ScgConstraintRow, lib.scg_sizeof_constraint_row), (ScgDisturbance, lib.scg_sizeof_disturbance),
                       (ScgTrajectorySink, lib.scg_sizeof_trajectory_sink)):
        if C.sizeof(struct) != fn():
            raise ScgError('%s layout mismatch: ctypes %d bytes, library %d bytes'
                           % (struct.__name__, C.sizeof(struct), fn()))
    _LIB = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().scg_last_error()
        raise ScgError('%s failed (status %d): %s' % (what, status, msg.decode() if msg else ''))


def current_stream(device):
    """Raw handle of torch's current HIP stream on ``device`` (the launches of the C ABI go there).  The raw accessor is
    what ``torch.cuda.current_stream()`` ends up calling, without ~4 us of Python in front of it."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)
    return C.c_void_p(raw(idx) if raw is not None else torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Device address of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
