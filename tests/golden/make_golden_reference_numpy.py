#!/usr/bin/env python
"""Generate tests/golden/reference_numpy.npz by EXECUTING the reference's own NumPy-only code for the hot path.

The reference's modules cannot be imported in the build container (casadi / pybullet / gym are absent: ordinary
ModuleNotFoundError), but many of the functions the path consists of need nothing but NumPy.  This script
parses the cited reference files with ``ast``, takes exactly the function / method / class definitions (or, where a method
mixes NumPy with PyBullet calls, a leading run of its statements) named in the ``SOURCES +=`` lines below, compiles them UNCHANGED
under their own file name and line numbers, and executes them with NumPy and plain attribute holders.  No module is
stubbed: nothing that needs casadi, pybullet or gym is executed (those functions stay "parity unpinned", DESIGN.md
section 2).  Two NumPy-1.x spellings the reference uses were removed in NumPy 2 and are aliased to their documented
equivalents in the namespace the code runs in: ``np.round_`` (= ``np.round``) and ``np.long`` (= ``int``).

Groups written (inputs and the reference's outputs, float64):
  motor_map      cmd2pwm / pwm2rpm                       envs/gym_pybullet_drones/quadrotor_utils.py:21-67
  trajectory     _generate_trajectory / _get_coordinates / _figure8 / _circle / _square   envs/benchmark_env.py:465-674
  transform      unit_vector / projection_matrix / transform_trajectory                   math_and_models/transformations.py:10-123
  x_goal         the X_GOAL / U_GOAL statements of Quadrotor.__init__                     envs/gym_pybullet_drones/quadrotor.py:261-329
  constraints    Constraint family, ConstraintList, create_constraint_list                envs/constraints.py:21-612
  step_glue      before_step / _preprocess_control / extend_obs / _get_done / _get_reward (rl) / the mse statements of
                 _get_info / after_step                   envs/benchmark_env.py:394-463, quadrotor.py:815-842,886-913,956-993,1011-1025
  gae            compute_returns_and_advantages          controllers/ppo/ppo_utils.py:428-456
  vec_env        DummyVecEnv (auto-reset, terminal stash) under VecRecordEpisodeStatistics (episode returns / lengths / trackers),
                 run over a deterministic toy env (tests/golden/toy_env.py)      envs/env_wrappers/vectorized_env/dummy_vec_env.py:10-59,
                                                                                 envs/env_wrappers/record_episode_statistics.py:92-169
  discretize     discretize_linear_system (exact and Euler; scipy.linalg.expm is importable)   controllers/mpc/mpc_utils.py:24-54
  normalization  RunningMeanStd, MeanStdNormalizer, RewardStdNormalizer                   math_and_models/normalization.py:17-190
  quadratic_cost the cost_func statement of Quadrotor._setup_symbolic on NumPy columns              envs/gym_pybullet_drones/quadrotor.py:682
  rk4            the Runge-Kutta statements of rk_discrete (stages k1 .. k4 and their combination), on NumPy vectors with the oracle's
                 continuous model as f                      controllers/mpc/mpc_utils.py:93-97
  vehicle        BaseAviary._parse_urdf_parameters on the reference's own assets/cf2x.urdf (xml.etree: stdlib), the "Compute constants"
                 statements of BaseAviary.__init__, and the NumPy statements of the explicit rigid-body model BaseAviary._dynamics
                 (forces, mixer torques, gyroscopic term, J_INV, world-frame acceleration), of _physics (forces, yaw torque), of
                 _ground_effect and of _drag, each cut before / after its PyBullet calls
                                                            envs/gym_pybullet_drones/base_aviary.py:100-147,412-423,437-496,522-558

Run in the build container only (the reference never travels to the GPU box):
    python tests/golden/make_golden_reference_numpy.py
tests/test_reference_pins.py replays the oracle (CPU) and the HIP path (GPU) against the file.
"""
import ast
import json
import math
import os
import sys
import types
import copy
from abc import ABC, abstractmethod
from collections import deque
from copy import deepcopy
from enum import Enum, IntEnum

import numpy as np
import scipy
import scipy.linalg

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = '/root/reference/safe_control_gym'


class NumpyCompat:
    """numpy, plus the two aliases of NumPy 1.x that NumPy 2 dropped and the reference still spells."""
    round_ = staticmethod(np.round)      # np.round_ was an alias of np.round (removed in NumPy 2.0)
    long = int                           # np.long was an alias of int (removed in NumPy 1.24)

    def __getattr__(self, name):
        return getattr(np, name)


NP = NumpyCompat()


# ---------------------------------------------------------------------------------------------------------------------
# AST extraction
# ---------------------------------------------------------------------------------------------------------------------
_TREES = {}


def _tree(rel):
    if rel not in _TREES:
        path = os.path.join(REF, rel)
        with open(path) as f:
            _TREES[rel] = (ast.parse(f.read(), filename=path), path)
    return _TREES[rel]


def _find(body, name):
    for node in body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name == name:
            return node
    raise KeyError(name)


def _exec_nodes(nodes, path, ns):
    mod = ast.Module(body=list(nodes), type_ignores=[])
    exec(compile(mod, path, 'exec'), ns)


def top_level(rel, names, ns):
    """Execute the top-level definitions `names` of a reference file (unchanged) in namespace `ns`."""
    tree, path = _tree(rel)
    _exec_nodes([_find(tree.body, n) for n in names], path, ns)
    return [(rel, n, _find(tree.body, n).lineno, _find(tree.body, n).end_lineno) for n in names]


def methods(rel, cls, names, ns):
    """The methods `names` of class `cls`, compiled unchanged as plain functions into `ns` (to be bound to a holder)."""
    tree, path = _tree(rel)
    c = _find(tree.body, cls)
    _exec_nodes([_find(c.body, n) for n in names], path, ns)
    return [(rel, '%s.%s' % (cls, n), _find(c.body, n).lineno, _find(c.body, n).end_lineno) for n in names]


def method_statements(rel, cls, name, pick, fn_name, ns, returns=None, body=None):
    """A run of statements of method `cls.name` selected by `pick(stmt) -> bool`, wrapped unchanged into a function
    ``fn_name(self)`` (for __init__ blocks, and for methods whose tail calls PyBullet or gym); `returns` names a local (or a
    tuple of locals) of those statements that the wrapper hands back (a method cut before its own ``return``); `body(method)`
    picks a nested statement list (a branch of an ``if``) instead of the method's top level."""
    tree, path = _tree(rel)
    m = _find(_find(tree.body, cls).body, name)
    chosen = [s for s in (body(m) if body else m.body) if pick(s)]
    assert chosen, (rel, cls, name)
    first, last = chosen[0].lineno, chosen[-1].end_lineno
    if returns:
        names = [returns] if isinstance(returns, str) else list(returns)
        value = ast.Name(id=names[0], ctx=ast.Load()) if len(names) == 1 else \
            ast.Tuple(elts=[ast.Name(id=n, ctx=ast.Load()) for n in names], ctx=ast.Load())
        chosen = chosen + [ast.Return(value=value)]
    fn = ast.FunctionDef(name=fn_name, args=ast.arguments(posonlyargs=[], args=[ast.arg(arg='self')], kwonlyargs=[],
                                                           kw_defaults=[], defaults=[]),
                         body=chosen, decorator_list=[], lineno=first, col_offset=0)
    ast.fix_missing_locations(fn)
    _exec_nodes([fn], path, ns)
    return [(rel, '%s.%s statements' % (cls, name), first, last)]


class Holder:
    """Plain attribute holder standing where the reference has `self` / `env`."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def bind(holder, ns, names):
    for n in names:
        setattr(holder, n, types.MethodType(ns[n], holder))


def _assigns_self(stmt, attr):
    return isinstance(stmt, ast.Assign) and any(isinstance(t, ast.Attribute) and t.attr == attr for t in stmt.targets)


def _tests_task(stmt):
    return isinstance(stmt, ast.If) and 'self.TASK' in ast.unparse(stmt.test) and 'X_GOAL' in ast.unparse(stmt)


# ---------------------------------------------------------------------------------------------------------------------
# the namespace every extracted piece runs in
# ---------------------------------------------------------------------------------------------------------------------
NS = {'np': NP, 'math': math, 'deepcopy': deepcopy, 'Enum': Enum, 'IntEnum': IntEnum, 'str': str, 'scipy': scipy,
      'copy': copy, 'ABC': ABC, 'abstractmethod': abstractmethod, 'deque': deque}
SOURCES = []
SOURCES += top_level('envs/gym_pybullet_drones/quadrotor_utils.py', ['QuadType', 'cmd2pwm', 'pwm2rpm'], NS)
SOURCES += top_level('envs/benchmark_env.py', ['Cost', 'Task'], NS)
SOURCES += top_level('math_and_models/transformations.py', ['unit_vector', 'projection_matrix', 'transform_trajectory'], NS)
SOURCES += top_level('envs/constraints.py', ['ConstrainedVariableType', 'Constraint', 'QuadraticContstraint', 'LinearConstraint',
                                             'BoundedConstraint', 'DefaultConstraint', 'SymmetricStateConstraint',
                                             'get_symbolic_constraint_models', 'ConstraintList', 'create_constraint_list'], NS)
SOURCES += top_level('controllers/ppo/ppo_utils.py', ['compute_returns_and_advantages', 'PPOBuffer', 'random_sample'], NS)
SOURCES += top_level('controllers/mpc/mpc_utils.py', ['discretize_linear_system'], NS)
SOURCES += top_level('envs/env_wrappers/vectorized_env/vec_env_utils.py', ['tile_images', '_flatten_obs', '_flatten_list'], NS)
SOURCES += top_level('envs/env_wrappers/vectorized_env/vec_env.py', ['VecEnv', 'VecEnvWrapper'], NS)
SOURCES += top_level('envs/env_wrappers/vectorized_env/dummy_vec_env.py', ['DummyVecEnv'], NS)
SOURCES += top_level('envs/env_wrappers/record_episode_statistics.py', ['VecRecordEpisodeStatistics'], NS)
SOURCES += top_level('math_and_models/normalization.py', ['RunningMeanStd', 'BaseNormalizer', 'MeanStdNormalizer',
                                                          'RewardStdNormalizer'], NS)
SOURCES += methods('envs/benchmark_env.py', 'BenchmarkEnv', ['_generate_trajectory', '_get_coordinates', '_figure8', '_circle',
                                                             '_square', 'before_step', 'extend_obs', 'after_step',
                                                             '_check_initial_reset', '_randomize_values_by_info'], NS)
SOURCES += methods('envs/gym_pybullet_drones/quadrotor.py', 'Quadrotor', ['_preprocess_control', '_get_reward', '_get_done'], NS)
SOURCES += method_statements('envs/gym_pybullet_drones/quadrotor.py', 'Quadrotor', '__init__',
                             lambda s: _assigns_self(s, 'U_GOAL') or _tests_task(s), 'build_goals', NS)
# _get_info: everything up to and including  info["mse"] = ...  (the rest of the method queries PyBullet contacts)
_seen_mse = []


def _until_mse(stmt):
    if _seen_mse:
        return False
    if isinstance(stmt, ast.Expr) and isinstance(stmt.value, ast.Constant):
        return False                                           # the docstring
    if '"mse"' in ast.unparse(stmt) or "'mse'" in ast.unparse(stmt):
        _seen_mse.append(1)
    return True


SOURCES += method_statements('envs/gym_pybullet_drones/quadrotor.py', 'Quadrotor', '_get_info', _until_mse, 'info_mse_part', NS,
                             returns='info')


# _set_observation_space: the thresholds and the low / high arrays of the state bounds -- everything before the first gym call
# (self.state_space = spaces.Box(low, high), which casts to float32)
_seen_box = []


def _until_state_space(stmt):
    if _seen_box or _assigns_self(stmt, 'state_space'):
        _seen_box.append(1)
        return False
    return not (isinstance(stmt, ast.Expr) and isinstance(stmt.value, ast.Constant))


SOURCES += method_statements('envs/gym_pybullet_drones/quadrotor.py', 'Quadrotor', '_set_observation_space', _until_state_space,
                             'state_bounds', NS, returns=('low', 'high'))


# _set_action_space: the two thrust-bound statements of its direct-thrust branch (the statements around them construct gym spaces)
def _direct_thrust_branch(method):
    branch = [s for s in method.body if isinstance(s, ast.If) and 'NORMALIZED_RL_ACTION_SPACE' in ast.unparse(s.test)][-1]
    return [ast.Assign(targets=[ast.Name(id='action_dim', ctx=ast.Store())],
                       value=ast.Attribute(value=ast.Name(id='self', ctx=ast.Load()), attr='action_dim_', ctx=ast.Load()),
                       lineno=branch.lineno)] + branch.orelse


def _thrust_bound(stmt):
    return isinstance(stmt, ast.Assign) and isinstance(stmt.targets[0], ast.Name) and stmt.targets[0].id in ('action_dim', 'n_motors', 'a_low', 'a_high')


SOURCES += method_statements('envs/gym_pybullet_drones/quadrotor.py', 'Quadrotor', '_set_action_space', _thrust_bound, 'thrust_bounds', NS,
                             returns=('a_low', 'a_high'), body=_direct_thrust_branch)


def GENERAL_CONSTRAINTS():
    return {'linear_constraint': NS['LinearConstraint'], 'quadratic_constraint': NS['QuadraticContstraint'],
            'bounded_constraint': NS['BoundedConstraint'], 'default_constraint': NS['DefaultConstraint'],
            'symmetric_state': NS['SymmetricStateConstraint']}


# -- the vehicle model as base_aviary.py states it in NumPy (everything between its PyBullet calls) ----------------------------------
import xml.etree.ElementTree as etxml                      # base_aviary.py:10 imports it under this name
NS['etxml'] = etxml
SOURCES += top_level('envs/gym_pybullet_drones/base_aviary.py', ['DroneModel'], NS)
SOURCES += methods('envs/gym_pybullet_drones/base_aviary.py', 'BaseAviary', ['_parse_urdf_parameters'], NS)


def function_statements(rel, name, pick, fn_name, ns, returns):
    """A run of statements of the top-level function `name`, wrapped unchanged into ``fn_name()`` (everything they read comes from
    the namespace) that returns the local `returns`: for functions whose head / tail build CasADi objects around plain arithmetic."""
    tree, path = _tree(rel)
    f = _find(tree.body, name)
    chosen = [st for st in f.body if pick(st)]
    assert chosen, (rel, name)
    first, last = chosen[0].lineno, chosen[-1].end_lineno
    fn = ast.FunctionDef(name=fn_name, args=ast.arguments(posonlyargs=[], args=[], kwonlyargs=[], kw_defaults=[], defaults=[]),
                         body=chosen + [ast.Return(value=ast.Name(id=returns, ctx=ast.Load()))], decorator_list=[], lineno=first, col_offset=0)
    ast.fix_missing_locations(fn)
    _exec_nodes([fn], path, ns)
    return [(rel, '%s statements' % name, first, last)]


def class_assignments(rel, cls, names, ns, prefix):
    """The class-level assignments `names` of class `cls` (its default tables), executed unchanged; the values land in `ns` under
    ``prefix + name``."""
    tree, path = _tree(rel)
    c = _find(tree.body, cls)
    out = []
    for n in names:
        node = next(st for st in c.body if isinstance(st, ast.Assign) and any(isinstance(t, ast.Name) and t.id == n for t in st.targets))
        scratch = dict(ns)
        _exec_nodes([node], path, scratch)
        ns[prefix + n] = scratch[n]
        out.append((rel, '%s.%s (class attribute)' % (cls, n), node.lineno, node.end_lineno))
    return out


# rk_discrete: the four stages and their combination (the CasADi symbols above them and the cs.Function below are not executed; X, U,
# f and dt are read from the namespace: NumPy vectors and a plain callable)
SOURCES += function_statements('controllers/mpc/mpc_utils.py', 'rk_discrete',
                                lambda st: isinstance(st, ast.Assign) and any(isinstance(t, ast.Name) and t.id in ('k1', 'k2', 'k3', 'k4', 'x_next')
                                                                              for t in st.targets), 'rk4_statements', NS, 'x_next')
# the quadratic cost of _setup_symbolic (quadrotor.py:682): one statement of operators on X, Xr, Q, U, Ur, R -- no cs.* call in it -- read
# from the namespace as NumPy column vectors / matrices (the CasADi symbols of the lines above it are not executed)
SOURCES += method_statements('envs/gym_pybullet_drones/quadrotor.py', 'Quadrotor', '_setup_symbolic',
                             lambda st: isinstance(st, ast.Assign) and any(isinstance(t, ast.Name) and t.id == 'cost_func' for t in st.targets),
                             'quadratic_cost_statement', NS, returns='cost_func')
SOURCES += class_assignments('envs/gym_pybullet_drones/quadrotor.py', 'Quadrotor',
                             ['DISTURBANCE_MODES', 'INERTIAL_PROP_RAND_INFO', 'INIT_STATE_RAND_INFO', 'TASK_INFO'], NS, 'Quadrotor_')
SOURCES += method_statements('envs/gym_pybullet_drones/quadrotor.py', 'Quadrotor', '__init__', lambda st: _assigns_self(st, 'INIT_STATE_LABELS'),
                             'init_state_labels', NS)


def _self_attr_assign(names):
    return lambda stmt: any(_assigns_self(stmt, a) for a in names)


SOURCES += method_statements('envs/gym_pybullet_drones/base_aviary.py', 'BaseAviary', '__init__',
                             _self_attr_assign(('GRAVITY', 'HOVER_RPM', 'MAX_RPM', 'MAX_THRUST', 'MAX_XY_TORQUE', 'MAX_Z_TORQUE',
                                                'GND_EFF_H_CLIP')), 'vehicle_constants', NS)


def _assigns_local(names):
    return lambda stmt: isinstance(stmt, ast.Assign) and any(isinstance(t, ast.Name) and t.id in names for t in stmt.targets)


# _dynamics: from `forces = ...` to `no_pybullet_dyn_accs = ...` (the state reads above use self.*, `rotation` comes from PyBullet; the
# statements below update the state and hand it to PyBullet).  `rpm`, `rotation`, `rpy_rates` are read as globals of the namespace.
SOURCES += method_statements('envs/gym_pybullet_drones/base_aviary.py', 'BaseAviary', '_dynamics',
                             lambda st: (isinstance(st, ast.If) and 'DRONE_MODEL' in ast.unparse(st.test)) or
                             _assigns_local(('forces', 'thrust', 'thrust_world_frame', 'force_world_frame', 'z_torques', 'z_torque',
                                             'torques', 'rpy_rates_deriv', 'no_pybullet_dyn_accs'))(st),
                             'explicit_dynamics', NS, returns=('no_pybullet_dyn_accs', 'rpy_rates_deriv'))
# _physics (the path Physics.PYB runs every sub-step): per-propeller forces and the yaw torque handed to PyBullet
SOURCES += method_statements('envs/gym_pybullet_drones/base_aviary.py', 'BaseAviary', '_physics', _assigns_local(('forces', 'torques', 'z_torque')),
                             'pyb_forces_and_yaw_torque', NS, returns=('forces', 'z_torque'))
SOURCES += method_statements('envs/gym_pybullet_drones/base_aviary.py', 'BaseAviary', '_ground_effect',
                             _assigns_local(('prop_heights', 'gnd_effects')), 'ground_effect_forces', NS, returns='gnd_effects')
SOURCES += method_statements('envs/gym_pybullet_drones/base_aviary.py', 'BaseAviary', '_drag', _assigns_local(('drag_factors', 'drag')),
                             'drag_force', NS, returns='drag')

# constants of the reference's vehicle (assets/cf2x.urdf:5,12-13; base_aviary.py:90,100-147)
CF = dict(MASS=0.03454, GRAVITY_ACC=9.8, KF=3.16e-10, KM=7.94e-12, PWM2RPM_SCALE=0.2685, PWM2RPM_CONST=4070.3, MIN_PWM=20000,
          MAX_PWM=65535)
F32MAX = float(np.finfo(np.float32).max)
OUT = {}
MANIFEST = {'generator': 'tests/golden/make_golden_reference_numpy.py', 'numpy': np.__version__,
            'sources': [dict(file='safe_control_gym/' + f, symbol=s, lines=[a, b]) for f, s, a, b in SOURCES], 'groups': {}}


def put(key, value):
    OUT[key] = np.asarray(value, dtype=np.float64)


def spaces(quad_type):
    """The state bounds as the reference's own statements build them (quadrotor.py:742-791, executed), cast to float32 as
    gym.spaces.Box(dtype=float32) holds them."""
    h = Holder(QUAD_TYPE=NS['QuadType'](quad_type), GROUND_PLANE_Z=0)          # base_aviary.py:121
    low, high = types.MethodType(NS['state_bounds'], h)()
    return Holder(low=np.asarray(low, dtype=np.float32), high=np.asarray(high, dtype=np.float32), shape=(len(low),))


def action_space(quad_type, normalized=False):
    """quadrotor.py:720-733: the direct-thrust bounds from the reference's own two statements (executed)."""
    nu = {1: 1, 2: 2, 3: 4}[quad_type]
    if normalized:
        return Holder(low=-np.ones(nu, dtype=np.float32), high=np.ones(nu, dtype=np.float32), shape=(nu,))
    h = Holder(action_dim_=nu, **CF)
    a_low, a_high = types.MethodType(NS['thrust_bounds'], h)()
    return Holder(low=np.full(nu, a_low, np.float32), high=np.full(nu, a_high, np.float32), shape=(nu,))


# ---------------------------------------------------------------------------------------------------------------------
def gen_spaces(rng):
    cases = []
    for qt in (1, 2, 3):
        sp, ac = spaces(qt), action_space(qt)
        put('spaces/q%d/state_low' % qt, sp.low)
        put('spaces/q%d/state_high' % qt, sp.high)
        put('spaces/q%d/thrust_low' % qt, ac.low)
        put('spaces/q%d/thrust_high' % qt, ac.high)
        cases.append('q%d' % qt)
    MANIFEST['groups']['spaces'] = dict(cases=cases, note='state bounds: the statements of Quadrotor._set_observation_space before its '
                                        'first gym call; thrust bounds: the a_low / a_high statements of _set_action_space; both cast '
                                        'to float32 as spaces.Box(dtype=np.float32) does; GROUND_PLANE_Z = 0 (base_aviary.py:121)')


def gen_motor_map(rng):
    cases = []
    for nu in (1, 2, 4):
        n_motor = 4 // nu
        lo = CF['KF'] * n_motor * (CF['PWM2RPM_SCALE'] * CF['MIN_PWM'] + CF['PWM2RPM_CONST']) ** 2
        hi = CF['KF'] * n_motor * (CF['PWM2RPM_SCALE'] * CF['MAX_PWM'] + CF['PWM2RPM_CONST']) ** 2
        thrusts = np.concatenate([rng.uniform(lo, hi, size=(12, nu)), rng.uniform(-0.1 * hi, 1.6 * hi, size=(8, nu)),
                                  np.zeros((1, nu)), np.full((1, nu), lo), np.full((1, nu), hi)])
        pwm = np.array([NS['cmd2pwm'](t.copy(), CF['PWM2RPM_SCALE'], CF['PWM2RPM_CONST'], CF['KF'], CF['MIN_PWM'], CF['MAX_PWM'])
                        for t in thrusts])
        rpm = np.array([NS['pwm2rpm'](p, CF['PWM2RPM_SCALE'], CF['PWM2RPM_CONST']) for p in pwm])
        put('motor_map/nu%d/thrust' % nu, thrusts)
        put('motor_map/nu%d/pwm' % nu, pwm)
        put('motor_map/nu%d/rpm' % nu, rpm)
        cases.append('nu%d' % nu)
    MANIFEST['groups']['motor_map'] = dict(cases=cases, constants=CF,
                                           note='pwm = cmd2pwm(thrust, ...) (4 motors), rpm = pwm2rpm(pwm, ...); the force of motor i '
                                                'is KF rpm_i^2 (base_aviary.py:420)')


TRAJ_CASES = [
    dict(traj_type='circle', traj_length=5.0, num_cycles=1, traj_plane='zx', position_offset=[0.5, 0.0], scaling=-0.5, sample_time=1 / 60),
    dict(traj_type='figure8', traj_length=5.0, num_cycles=1, traj_plane='xy', position_offset=[0.0, 0.0], scaling=1.0, sample_time=1 / 50),
    dict(traj_type='square', traj_length=6.0, num_cycles=2, traj_plane='xz', position_offset=[0.2, 1.0], scaling=0.8, sample_time=1 / 30),
    dict(traj_type='figure8', traj_length=4.0, num_cycles=2, traj_plane='yz', position_offset=[-0.3, 0.7], scaling=0.35, sample_time=1 / 60),
    dict(traj_type='circle', traj_length=3.0, num_cycles=3, traj_plane='yx', position_offset=[0.0, 0.1], scaling=1.5, sample_time=0.01),
    dict(traj_type='square', traj_length=5.0, num_cycles=1, traj_plane='zy', position_offset=[1.0, 0.0], scaling=-0.4, sample_time=1 / 60),
]


def gen_trajectory():
    h = Holder()
    bind(h, NS, ['_generate_trajectory', '_get_coordinates', '_figure8', '_circle', '_square'])
    for k, c in enumerate(TRAJ_CASES):
        kw = dict(c, position_offset=np.array(c['position_offset'], dtype=float))
        pos, vel, speed = h._generate_trajectory(**kw)
        put('trajectory/%d/pos' % k, pos)
        put('trajectory/%d/vel' % k, vel)
        put('trajectory/%d/speed' % k, speed)
    MANIFEST['groups']['trajectory'] = dict(cases=TRAJ_CASES)


TRANSFORM_CASES = [dict(point=[0.0, 0.0, 0.5], normal=[0.0, 1.0, 1.0]), dict(point=[0.3, -0.2, 1.0], normal=[1.0, 0.0, 0.0]),
                   dict(point=[0.0, 0.0, 0.0], normal=[0.2, -0.7, 0.4]), dict(point=[1.0, 2.0, 3.0], normal=[0.0, 0.0, -2.0])]


def gen_transform(rng):
    for k, c in enumerate(TRANSFORM_CASES):
        # ndarray arguments: projection_matrix says np.array(point[:3], dtype=float64, copy=False), which NumPy 2 refuses for a
        # list (a copy is unavoidable) but accepts for a float64 array, the same values either way
        point, normal = np.array(c['point'], dtype=np.float64), np.array(c['normal'], dtype=np.float64)
        put('transform/%d/unit_normal' % k, NS['unit_vector'](normal))
        put('transform/%d/M' % k, NS['projection_matrix'](point, normal))
        pos, vel = rng.normal(size=(7, 3)), rng.normal(size=(7, 3))
        tp, tv = NS['transform_trajectory'](pos, vel, trans_info={'point': point, 'normal': normal})
        put('transform/%d/pos' % k, pos)
        put('transform/%d/vel' % k, vel)
        put('transform/%d/trans_pos' % k, tp)
        put('transform/%d/trans_vel' % k, tv)
    MANIFEST['groups']['transform'] = dict(cases=TRANSFORM_CASES,
                                           note='transform_trajectory augments the VELOCITY rows with ones as well '
                                                '(transformations.py:121): the plane offset M[:3, 3] is added to the velocities')


DEFAULT_TASK_INFO = {'stabilization_goal': [0, 1], 'stabilization_goal_tolerance': 0.05, 'trajectory_type': 'circle', 'num_cycles': 1,
                     'trajectory_plane': 'zx', 'trajectory_position_offset': [0.5, 0], 'trajectory_scale': -0.5,
                     'proj_point': [0, 0, 0.5], 'proj_normal': [0, 1, 1]}           # quadrotor.py:136-146
X_GOAL_CASES = [
    dict(quad_type=3, task='traj_tracking', ctrl_freq=60, episode_len_sec=5, task_info={}),
    dict(quad_type=3, task='traj_tracking', ctrl_freq=50, episode_len_sec=4,
         task_info={'trajectory_type': 'figure8', 'trajectory_plane': 'xy', 'trajectory_scale': 0.75, 'num_cycles': 2,
                    'trajectory_position_offset': [0.1, -0.2], 'proj_point': [0, 0, 1.0], 'proj_normal': [0, 0, 1]}),
    dict(quad_type=3, task='stabilization', ctrl_freq=60, episode_len_sec=5, task_info={'stabilization_goal': [0.5, -0.5, 1.2]}),
    dict(quad_type=2, task='traj_tracking', ctrl_freq=60, episode_len_sec=5, task_info={}),
    dict(quad_type=2, task='traj_tracking', ctrl_freq=30, episode_len_sec=6,
         task_info={'trajectory_type': 'square', 'trajectory_plane': 'xz', 'trajectory_scale': 0.6, 'num_cycles': 2}),
    dict(quad_type=2, task='stabilization', ctrl_freq=60, episode_len_sec=5, task_info={}),
    dict(quad_type=1, task='traj_tracking', ctrl_freq=60, episode_len_sec=5, task_info={'trajectory_type': 'figure8'}),
    dict(quad_type=1, task='stabilization', ctrl_freq=60, episode_len_sec=5, task_info={}),
]


def goal_holder(case):
    """`self` for the X_GOAL / U_GOAL statements of Quadrotor.__init__ (quadrotor.py:261-329)."""
    ti = dict(DEFAULT_TASK_INFO)
    ti.update(case['task_info'])
    ti['proj_point'] = np.array(ti['proj_point'], dtype=np.float64)        # see gen_transform: ndarray for NumPy 2
    ti['proj_normal'] = np.array(ti['proj_normal'], dtype=np.float64)
    ti['trajectory_position_offset'] = np.array(ti['trajectory_position_offset'], dtype=float)
    nx, nu = {1: (2, 1), 2: (6, 2), 3: (12, 4)}[case['quad_type']]
    h = Holder(QUAD_TYPE=NS['QuadType'](case['quad_type']), TASK=NS['Task'](case['task']), TASK_INFO=ti, MASS=CF['MASS'],
               GRAVITY_ACC=CF['GRAVITY_ACC'], action_dim=nu, state_dim=nx, EPISODE_LEN_SEC=case['episode_len_sec'],
               CTRL_TIMESTEP=1.0 / case['ctrl_freq'])
    bind(h, NS, ['_generate_trajectory', '_get_coordinates', '_figure8', '_circle', '_square', 'build_goals'])
    h.build_goals()
    return h


def gen_x_goal():
    for k, c in enumerate(X_GOAL_CASES):
        h = goal_holder(c)
        put('x_goal/%d/X_GOAL' % k, h.X_GOAL)
        put('x_goal/%d/U_GOAL' % k, h.U_GOAL)
    MANIFEST['groups']['x_goal'] = dict(cases=X_GOAL_CASES, default_task_info=DEFAULT_TASK_INFO)


# ---- constraints -----------------------------------------------------------------------------------------------------
CONSTRAINT_CASES = [
    dict(name='competition_3d', quad_type=3, specs=[
        {'constraint_form': 'default_constraint', 'constrained_variable': 'input'},
        {'constraint_form': 'bounded_constraint', 'constrained_variable': 'state', 'active_dims': [0, 2, 4],
         'upper_bounds': [3, 3, 2], 'lower_bounds': [-3, -3, -0.1]}]),                        # competition/getting_started.yaml:123-156
    dict(name='bench_c3', quad_type=2, specs=[
        {'constraint_form': 'bounded_constraint', 'constrained_variable': 'state', 'active_dims': [0, 2, 4],
         'lower_bounds': [-2, 0, -0.5], 'upper_bounds': [2, 2, 0.5]}]),
    dict(name='default_state_and_input_2d', quad_type=2, specs=[
        {'constraint_form': 'default_constraint', 'constrained_variable': 'input'},
        {'constraint_form': 'default_constraint', 'constrained_variable': 'state'}]),
    dict(name='linear_general_strict_tol', quad_type=2, specs=[
        {'constraint_form': 'linear_constraint', 'constrained_variable': 'state', 'strict': True, 'tolerance': [0.1, 0.2],
         'A': [[1.0, 0, 0.5, 0, 0, 0], [0, 0, -1.0, 0, 2.0, 0]], 'b': [1.0, 0.3]},
        {'constraint_form': 'linear_constraint', 'constrained_variable': 'input', 'A': [[1.0, -1.0]], 'b': [0.05]}]),
    dict(name='quadratic_ball', quad_type=3, specs=[
        {'constraint_form': 'quadratic_constraint', 'constrained_variable': 'state', 'active_dims': [0, 2, 4],
         'P': [[1.0, 0.1, 0], [0.1, 2.0, 0], [0, 0, 0.5]], 'b': 1.5, 'tolerance': [0.25]},
        {'constraint_form': 'bounded_constraint', 'constrained_variable': 'input', 'lower_bounds': [0.03, 0.03, 0.03, 0.03],
         'upper_bounds': [0.14, 0.14, 0.14, 0.14], 'tolerance': [0.01] * 8}]),
    dict(name='bounded_1d_tol', quad_type=1, specs=[
        {'constraint_form': 'bounded_constraint', 'constrained_variable': 'state', 'lower_bounds': [0.2, -1.0],
         'upper_bounds': [1.8, 1.0], 'tolerance': [0.05, 0.05, 0.05, 0.05], 'strict': True}]),
    dict(name='cartpole_symmetric', quad_type='cartpole', specs=[
        {'constraint_form': 'symmetric_state', 'constrained_variable': 'state', 'active_dims': [0, 2], 'bound': [1.0, 0.2],
         'tolerance': [0.1, 0.02]}]),
]


def constraint_env(case):
    qt = case['quad_type']
    if qt == 'cartpole':
        nx, nu = 4, 1
        ss = Holder(low=np.array([-4.8, -F32MAX, -math.pi, -F32MAX], dtype=np.float32),
                    high=np.array([4.8, F32MAX, math.pi, F32MAX], dtype=np.float32), shape=(4,))
        asp = Holder(low=np.array([-10.0], dtype=np.float32), high=np.array([10.0], dtype=np.float32), shape=(1,))
        name = 'cartpole'
    else:
        nx, nu = {1: (2, 1), 2: (6, 2), 3: (12, 4)}[qt]
        ss, asp, name = spaces(qt), action_space(qt), 'quadrotor'
    return Holder(state_dim=nx, action_dim=nu, state_space=ss, observation_space=ss, action_space=asp, NAME=name, COST='rl_reward',
                  state=np.zeros(nx), current_raw_input_action=np.zeros(nu))


def sample_states(rng, env, count):
    nx, nu = env.state_dim, env.action_dim
    lo = np.where(env.state_space.low < -1e30, -3.0, env.state_space.low).astype(float)
    hi = np.where(env.state_space.high > 1e30, 3.0, env.state_space.high).astype(float)
    xs = rng.uniform(lo - 0.4 * (hi - lo), hi + 0.4 * (hi - lo), size=(count, nx))
    xs[0] = 0.0
    al, ah = env.action_space.low.astype(float), env.action_space.high.astype(float)
    us = rng.uniform(al - 0.3 * (ah - al), ah + 0.3 * (ah - al), size=(count, nu))
    return xs, us


def gen_constraints(rng):
    for c in CONSTRAINT_CASES:
        env = constraint_env(c)
        specs = deepcopy(c['specs'])
        for s in specs:                                   # the yaml loader hands lists; the classes take them as they are
            if s['constraint_form'] == 'symmetric_state':
                s['bound'] = np.array(s['bound'], dtype=float)     # `-bound` in the constructor needs an array (constraints.py:393)
        cl = NS['create_constraint_list'](specs, GENERAL_CONSTRAINTS(), env)
        xs, us = sample_states(rng, env, 24)
        vals, vals_state, viol, almost, per_viol = [], [], [], [], []
        for x, u in zip(xs, us):
            env.state, env.current_raw_input_action = x.copy(), u.copy()
            v = cl.get_values(env)
            vals.append(v)
            viol.append(cl.is_violated(env, c_value=v))
            almost.append(cl.is_almost_active(env, c_value=v))
            per_viol.append(cl.get_violations(env))
            if cl.state_constraints:
                vals_state.append(cl.get_values(env, only_state=True))
        key = 'constraints/%s/' % c['name']
        put(key + 'state', xs)
        put(key + 'action', us)
        put(key + 'values', vals)
        put(key + 'violated', viol)
        put(key + 'almost_active', almost)
        put(key + 'violations_per_constraint', per_viol)
        if vals_state:
            put(key + 'values_only_state', vals_state)
        put(key + 'num_constraints', [cl.num_constraints])
    MANIFEST['groups']['constraints'] = dict(cases=CONSTRAINT_CASES,
                                             note='values = ConstraintList.get_values(env) (rounded to 8 decimals by the reference, '
                                                  'constraints.py:119); violated / almost_active evaluated on those values')


# ---- step glue -------------------------------------------------------------------------------------------------------
GLUE_CASES = [
    dict(name='q3_track_rl', quad_type=3, task='traj_tracking', cost='rl_reward', obs_goal_horizon=2),
    dict(name='q3_stab_rl_con_penalty', quad_type=3, task='stabilization', cost='rl_reward', obs_goal_horizon=1,
         task_info={'stabilization_goal': [0.2, -0.1, 1.0]}, constraints='competition_3d', use_constraint_penalty=True,
         constraint_penalty=-2.5, rew_exponential=False, rew_state_weight=[1, 0.1, 1, 0.1, 2, 0.1, 0.5, 0.5, 0.5, 0.01, 0.01, 0.01],
         rew_act_weight=0.3),
    dict(name='q2_stab_quadratic_done', quad_type=2, task='stabilization', cost='quadratic'),
    dict(name='q2_track_rl_norm_act_done_on_violation', quad_type=2, task='traj_tracking', cost='rl_reward',
         normalized_rl_action_space=True, constraints='bench_c3', done_on_violation=True, obs_goal_horizon=0),
    dict(name='q1_stab_rl', quad_type=1, task='stabilization', cost='rl_reward', obs_goal_horizon=1, done_on_out_of_bound=False),
    dict(name='q1_track_rl_bounded_tol', quad_type=1, task='traj_tracking', cost='rl_reward', constraints='bounded_1d_tol',
         use_constraint_penalty=True, constraint_penalty=-1.0),
]


def glue_env(c):
    qt = c['quad_type']
    gh = goal_holder(dict(quad_type=qt, task=c['task'], ctrl_freq=60, episode_len_sec=5, task_info=c.get('task_info', {})))
    nx, nu = gh.state_dim, gh.action_dim
    norm = c.get('normalized_rl_action_space', False)
    mse_w = {1: [1, 0], 2: [1, 0, 1, 0, 0, 0], 3: [1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0]}[qt]
    h = Holder(QUAD_TYPE=gh.QUAD_TYPE, TASK=gh.TASK, COST=NS['Cost'](c['cost']), TASK_INFO=gh.TASK_INFO, X_GOAL=gh.X_GOAL, U_GOAL=gh.U_GOAL,
               state_dim=nx, action_dim=nu, state_space=spaces(qt), observation_space=spaces(qt), action_space=action_space(qt, norm),
               NAME='quadrotor', NORMALIZED_RL_ACTION_SPACE=norm, norm_act_scale=0.1, hover_thrust=CF['GRAVITY_ACC'] * CF['MASS'] / nu,
               VERBOSE=False, disturbances={}, adversary_disturbance=None, initial_reset=True,
               obs_goal_horizon=c.get('obs_goal_horizon', 0),
               rew_state_weight=np.array(c.get('rew_state_weight', 1.0), ndmin=1, dtype=float),
               rew_act_weight=np.array(c.get('rew_act_weight', 0.0001), ndmin=1, dtype=float),
               rew_exponential=c.get('rew_exponential', True), done_on_out_of_bound=c.get('done_on_out_of_bound', True),
               info_mse_metric_state_weight=np.array(mse_w, ndmin=1, dtype=float),
               DONE_ON_COLLISION=False, DONE_ON_COMPLETION=False, currently_collided=False, task_completed=False,
               DONE_ON_VIOLATION=c.get('done_on_violation', False), use_constraint_penalty=c.get('use_constraint_penalty', False),
               constraint_penalty=c.get('constraint_penalty', -1), PYB_STEPS_PER_CTRL=4, CTRL_STEPS=300, CTRL_FREQ=60,
               constraints=None, goal_reached=False, **{k: CF[k] for k in ('PWM2RPM_SCALE', 'PWM2RPM_CONST', 'KF', 'MIN_PWM', 'MAX_PWM')})
    bind(h, NS, ['before_step', '_check_initial_reset', '_preprocess_control', 'extend_obs', 'after_step', '_get_reward', '_get_done',
                 'info_mse_part'])
    if c.get('constraints'):
        spec = [cc for cc in CONSTRAINT_CASES if cc['name'] == c['constraints']][0]
        h.constraints = NS['create_constraint_list'](deepcopy(spec['specs']), GENERAL_CONSTRAINTS(), h)
    return h


def gen_step_glue(rng):
    for c in GLUE_CASES:
        h = glue_env(c)
        nx, nu = h.state_dim, h.action_dim
        rec = {k: [] for k in ('state', 'action', 'counter', 'rpm', 'clipped', 'obs', 'mse', 'done_raw', 'goal_reached', 'rew_raw', 'rew',
                               'done', 'violation', 'truncated', 'cvals')}
        tracking = h.TASK == NS['Task'].TRAJ_TRACKING
        for k in range(40):
            counter = int(rng.choice([0, 1, 7, 150, 298, 299])) if k % 4 else int(rng.integers(0, 299))
            goal = h.X_GOAL[min(counter, h.X_GOAL.shape[0] - 1)] if tracking else h.X_GOAL
            spread = [0.02, 0.3, 2.0, 6.0][k % 4]                  # near the goal ... far out of bounds
            x = goal + spread * rng.normal(size=nx)
            al, ah = h.action_space.low.astype(float), h.action_space.high.astype(float)
            a = rng.uniform(al - 0.4 * (ah - al), ah + 0.4 * (ah - al))
            # ---- the reference's step() order (quadrotor.py:507-558) with the state advance replaced by "state := x"
            h.ctrl_step_counter, h.pyb_step_counter = counter, 4 * counter
            rpm = h.before_step(a.copy())                          # -> _preprocess_control
            h.state = x.copy()                                     # stands for _advance_simulation + _get_observation's self.state
            obs = h.extend_obs(deepcopy(h.state), h.ctrl_step_counter + 1)          # quadrotor.py:883
            info = h.info_mse_part()                               # the statements of _get_info up to info["mse"]
            done = h._get_done()
            goal_reached = bool(getattr(h, 'goal_reached', False))
            rew = h._get_reward() if h.COST == NS['Cost'].RL_REWARD else float('nan')     # quadratic cost: CasADi (unpinned)
            rew_in = rew if rew == rew else 0.0
            obs2, rew2, done2, info2 = h.after_step(obs, rew_in, done, dict(info))
            rec['state'].append(x); rec['action'].append(a); rec['counter'].append(counter); rec['rpm'].append(rpm)
            rec['clipped'].append(np.asarray(h.current_preprocessed_action, dtype=float)); rec['obs'].append(obs2)
            rec['mse'].append(info['mse']); rec['done_raw'].append(done); rec['goal_reached'].append(goal_reached)
            rec['rew_raw'].append(rew); rec['rew'].append(rew2 if rew == rew else float('nan')); rec['done'].append(done2)
            rec['violation'].append(info2.get('constraint_violation', -1))
            rec['truncated'].append({True: 1, False: 0}.get(info2.get('TimeLimit.truncated'), -1))
            rec['cvals'].append(info2.get('constraint_values', np.zeros(0)))
        for k, v in rec.items():
            put('step_glue/%s/%s' % (c['name'], k), v)
    MANIFEST['groups']['step_glue'] = dict(
        cases=GLUE_CASES,
        note='per sample: before_step(action) -> rpm and current_preprocessed_action (clipped); state := the given state (standing for '
             'the PyBullet advance); obs = extend_obs(state, counter + 1); mse = the statements of _get_info up to info["mse"]; '
             'done_raw = _get_done(); rew_raw = _get_reward() (rl_reward only: the quadratic cost calls CasADi, NaN here); then '
             'after_step(obs, rew, done, info) -> rew, done, constraint_values / violation (-1: no constraints), '
             'TimeLimit.truncated (-1: key absent).  counter = ctrl_step_counter before after_step increments it.')


# ---- step glue on states the engine can reach ------------------------------------------------------------------------
def gen_step_glue_device(rng):
    """The same executed reference methods as gen_step_glue, but the state handed to them is one control step of the prior
    model away from a stored (x0, action): x = rk4(x0, thrust(action)) with the oracle's float64 RK4 and the thrusts the
    REFERENCE's before_step / _preprocess_control produced (rpm -> KF rpm^2).  A device test can therefore start the HIP env at
    x0, step it with the action and compare reward / done / truncation / violation / constraint values / observation with
    what the reference's own _get_done / _get_reward / after_step say about that step -- a direct pin of the fused kernel's
    glue instead of a transitive one.  Samples within 1e-3 of any decision boundary (out-of-bound box, constraint rows and their
    tolerances, time limit is exact) are rejected, so that a 1e-6 difference of the state cannot flip a recorded flag."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import ref_model as rm
    dt = 1.0 / 60
    for c in GLUE_CASES:
        h = glue_env(c)
        nx, nu = h.state_dim, h.action_dim
        f = rm.F_BY_SYSTEM[c['quad_type']]
        params = rm.QuadParams()
        rec = {k: [] for k in ('state0', 'action', 'counter', 'state', 'obs', 'mse', 'rew', 'done', 'violation', 'truncated', 'cvals',
                               'goal_reached')}
        tracking = h.TASK == NS['Task'].TRAJ_TRACKING
        lo, hi = h.state_space.low.astype(float), h.state_space.high.astype(float)
        kept = tries = 0
        while kept < 32:
            tries += 1
            assert tries < 4000, c['name']
            counter = int(rng.choice([0, 1, 7, 150, 297, 298, 299])) if kept % 3 else int(rng.integers(0, 299))
            goal = h.X_GOAL[min(counter, h.X_GOAL.shape[0] - 1)] if tracking else h.X_GOAL
            spread = [0.02, 0.3, 1.5][kept % 3]
            x0 = (goal + spread * rng.normal(size=nx)).astype(np.float32).astype(float)      # what the device will hold
            al, ah = h.action_space.low.astype(float), h.action_space.high.astype(float)
            a = rng.uniform(al - 0.4 * (ah - al), ah + 0.4 * (ah - al)).astype(np.float32).astype(float)
            h.ctrl_step_counter, h.pyb_step_counter = counter, 4 * counter
            rpm = h.before_step(a.copy())
            kf = CF['KF'] * np.asarray(rpm, dtype=float) ** 2                                # base_aviary.py:420, per motor
            thrust = [kf.sum()] if nu == 1 else ([kf[0] + kf[3], kf[1] + kf[2]] if nu == 2 else list(kf))
            x = np.array(rm.rk4_step(f, list(x0), thrust, params, dt))
            if not np.all(np.isfinite(x)) or np.any(np.abs(x - lo) < 1e-3) or np.any(np.abs(x - hi) < 1e-3):
                continue
            if c['quad_type'] == 3 and abs(x[7]) > 1.3:                                      # the Euler-rate matrix is ill-conditioned there
                continue
            h.state = x.copy()
            obs = h.extend_obs(deepcopy(h.state), h.ctrl_step_counter + 1)
            info = h.info_mse_part()
            done = h._get_done()
            goal_reached = bool(getattr(h, 'goal_reached', False))
            if c['cost'] == 'quadratic' and not tracking and abs(np.linalg.norm(x - goal) - h.TASK_INFO['stabilization_goal_tolerance']) < 1e-3:
                continue
            rew = h._get_reward() if h.COST == NS['Cost'].RL_REWARD else float('nan')
            obs2, rew2, done2, info2 = h.after_step(obs, rew if rew == rew else 0.0, done, dict(info))
            cv = np.asarray(info2.get('constraint_values', np.zeros(0)), dtype=float)
            if cv.size and (np.any(np.abs(cv) < 1e-3) or
                            (h.use_constraint_penalty and np.any(np.abs(cv + np.concatenate([np.atleast_1d(con.tolerance if con.tolerance is not None else np.full(con.num_constraints, np.inf))
                                                                                     for con in h.constraints.constraints])) < 1e-3))):
                continue
            rec['state0'].append(x0); rec['action'].append(a); rec['counter'].append(counter); rec['state'].append(x)
            rec['obs'].append(obs2); rec['mse'].append(info['mse']); rec['goal_reached'].append(goal_reached)
            rec['rew'].append(rew2 if rew == rew else float('nan')); rec['done'].append(done2)
            rec['violation'].append(info2.get('constraint_violation', -1))
            rec['truncated'].append({True: 1, False: 0}.get(info2.get('TimeLimit.truncated'), -1))
            rec['cvals'].append(cv)
            kept += 1
        for k, v in rec.items():
            put('step_glue_device/%s/%s' % (c['name'], k), v)
    MANIFEST['groups']['step_glue_device'] = dict(
        cases=GLUE_CASES, dt=dt,
        note='per sample: state0, action (both float32-representable), counter; state = oracle float64 RK4 step of state0 under the '
             'thrusts KF rpm^2 of the reference-executed before_step(action) (the one ingredient that is not the reference: the '
             'dynamics are CasADi, parity unpinned); then the reference-executed extend_obs / mse statements / _get_done / '
             '_get_reward (rl) / after_step on that state, as in step_glue.  Samples within 1e-3 of a decision boundary were rejected.')


# ---- GAE and normalisers ---------------------------------------------------------------------------------------------
def gen_gae(rng):
    cases = []
    for k, (T, N, use_gae, with_term) in enumerate([(8, 5, True, True), (8, 5, False, True), (16, 3, True, False), (1, 4, True, True),
                                                    (33, 7, True, True)]):
        rews = rng.normal(size=(T, N, 1))
        vals = rng.normal(size=(T, N, 1))
        masks = (rng.uniform(size=(T, N, 1)) > 0.2).astype(float)
        last_val = rng.normal(size=(N, 1))
        term = (rng.normal(size=(T, N, 1)) * (masks == 0)) if with_term else 0
        gamma, lam = 0.97 + 0.005 * k, 0.9 + 0.01 * k
        rews_in = rews.copy()
        rets, advs = NS['compute_returns_and_advantages'](rews_in, vals.copy(), masks.copy(), deepcopy(term), last_val.copy(), gamma=gamma,
                                                           use_gae=use_gae, gae_lambda=lam)
        key = 'gae/%d/' % k
        put(key + 'rews', rews); put(key + 'vals', vals); put(key + 'masks', masks); put(key + 'last_val', last_val)
        put(key + 'terminal_vals', term if with_term else np.zeros((T, N, 1)))
        put(key + 'rews_after', rews_in)                      # the reference updates rews in place (:443)
        put(key + 'rets', rets); put(key + 'advs', advs)
        cases.append(dict(T=T, N=N, use_gae=use_gae, with_terminal_vals=with_term, gamma=gamma, gae_lambda=lam))
    MANIFEST['groups']['gae'] = dict(cases=cases)


def gen_discretize(rng):
    cases = []
    for k, (nx, nu, norm, dt) in enumerate([(12, 4, 8.0, 1 / 60), (6, 2, 40.0, 1 / 50), (4, 1, 2.0, 0.02), (2, 1, 300.0, 1 / 60),
                                            (12, 4, 900.0, 1 / 30)]):
        A = rng.normal(size=(6, nx, nx))
        A *= norm / np.abs(A).sum(2).max(1)[:, None, None]
        B = rng.normal(size=(6, nx, nu)) * [[[0.5]], [[30.0]], [[2000.0]], [[1.0]], [[5.0]], [[100.0]]]
        for exact in (True, False):
            out = [NS['discretize_linear_system'](A[i], B[i], dt, exact=exact) for i in range(6)]
            put('discretize/%d/Ad_%s' % (k, 'exact' if exact else 'euler'), [o[0] for o in out])
            put('discretize/%d/Bd_%s' % (k, 'exact' if exact else 'euler'), [o[1] for o in out])
        put('discretize/%d/A' % k, A)
        put('discretize/%d/B' % k, B)
        cases.append(dict(nx=nx, nu=nu, dt=dt, norm_inf_A=norm))
    MANIFEST['groups']['discretize'] = dict(cases=cases)


RAND_SPECS = {
    'a_normal_kw': {'distrib': 'normal', 'loc': 0.1, 'scale': 0.2}, 'b_normal_args': {'distrib': 'normal', 'args': [-0.3], 'scale': 0.05},
    'c_standard_normal': {'distrib': 'standard_normal'}, 'd_exponential': {'distrib': 'exponential', 'scale': 0.3},
    'e_laplace': {'distrib': 'laplace', 'loc': 0.2, 'scale': 0.1}, 'f_logistic': {'distrib': 'logistic', 'args': [0.0, 0.05]},
    'g_triangular': {'distrib': 'triangular', 'left': -0.2, 'mode': 0.1, 'right': 0.3}, 'h_randint': {'distrib': 'randint', 'low': -2, 'high': 3},
    'i_randint_one_arg': {'distrib': 'randint', 'args': [4]}, 'j_uniform': {'distrib': 'uniform', 'low': -0.05, 'high': 0.15},
    'k_uniform_args': {'distrib': 'uniform', 'args': [-0.5], 'high': 0.25},
}


def gen_randomization(rng):
    """BenchmarkEnv._randomize_values_by_info (benchmark_env.py:277-308) with numpy.random.RandomState standing where gym's
    seeding.np_random puts one (gym 0.21: a RandomState): value = original + np_random.<distrib>(*args, **kwargs), keys visited
    in the order of `original_values`, keys without an info entry untouched, the info dict not consumed."""
    original = {k: 0.25 * (i + 1) for i, k in enumerate(sorted(RAND_SPECS))}
    original['z_untouched'] = 7.0
    info = copy.deepcopy(RAND_SPECS)
    h = Holder(np_random=np.random.RandomState(2024))
    bind(h, NS, ['_randomize_values_by_info'])
    rows = []
    for _ in range(48):
        out = h._randomize_values_by_info(original, info)
        rows.append([out[k] for k in sorted(original)])
    assert info == RAND_SPECS, 'the reference consumed the caller\'s info dict'
    put('randomization/values', np.asarray(rows, dtype=np.float64))
    put('randomization/original', [original[k] for k in sorted(original)])
    MANIFEST['groups']['randomization'] = dict(keys=sorted(original), specs=RAND_SPECS, seed=2024, calls=48,
                                               note='rows = successive calls with one RandomState(seed); columns = sorted keys')


def gen_ppo_buffer(rng):
    """PPOBuffer (ppo_utils.py:282-407) and random_sample (:410-426).  `Box` is only the right-hand side of one isinstance():
    bound to an empty class, the action space is then read through `.n` (the reference's own second branch)."""
    import torch
    NS['Box'] = type('Box', (), {})
    NS['torch'] = torch
    T, N, od, ad = 5, 3, 4, 2
    buf = NS['PPOBuffer'](Holder(shape=(od,)), Holder(n=ad), T, N)
    pushes = []
    for k in range(7):                                                  # two more than max_length: the ring wraps
        b = {'obs': rng.normal(size=(N, od)), 'act': rng.normal(size=(N, ad)), 'rew': rng.normal(size=N), 'mask': (rng.uniform(size=N) > 0.3) * 1.0,
             'v': rng.normal(size=(N, 1)), 'logp': rng.normal(size=N), 'terminal_v': rng.normal(size=N)}
        if k == 3:
            b.pop('logp')                                               # a partial push leaves the other containers alone
        pushes.append(b)
        buf.push(b)
        for key, v in b.items():
            put('ppo_buffer/push%d/%s' % (k, key), v)
    put('ppo_buffer/t', buf.t)
    got = buf.get()
    for key in buf.keys:
        put('ppo_buffer/get/%s' % key, got[key].numpy())
    idx = np.array([0, 14, 7, 3, 11])
    smp = buf.sample(idx)
    put('ppo_buffer/sample_indices', idx)
    for key in buf.keys:
        put('ppo_buffer/sample/%s' % key, smp[key])
    sizes = {}
    for drop_last in (True, False):
        np.random.seed(5)
        batches = [b['obs'].shape[0] for b in buf.sampler(4, drop_last=drop_last)]
        np.random.seed(5)
        ids = np.concatenate(list(NS['random_sample'](np.arange(T * N), 4, drop_last)))
        sizes[str(drop_last)] = batches
        put('ppo_buffer/sampler_indices_drop_last_%s' % drop_last, ids)
    MANIFEST['groups']['ppo_buffer'] = dict(T=T, N=N, obs_dim=od, act_dim=ad, pushes=7, keys=list(buf.keys), sampler_batch_sizes=sizes,
                                            push_keys=[sorted(b.keys()) for b in pushes])


def gen_vec_env(rng):
    from toy_env import ToyEnv
    horizons = [3, 5, 4, 7]
    venv = NS['VecRecordEpisodeStatistics'](NS['DummyVecEnv']([(lambda i=i, h=h: ToyEnv(i, h)) for i, h in enumerate(horizons)]), deque_size=50)
    venv.add_tracker('constraint_violation', 0, mode='queue')
    obs0, _ = venv.reset()
    put('vec_env/obs0', obs0)
    T = 30
    actions = rng.uniform(-1.0, 1.0, size=(T, len(horizons), 1))
    obs_l, rew_l, done_l, term_l, epr_l, epl_l, viol_l = [], [], [], [], [], [], []
    for t in range(T):
        obs, rew, done, info = venv.step(actions[t])
        obs_l.append(obs); rew_l.append(rew); done_l.append(done)
        term_l.append([info['n'][i]['terminal_observation'] if done[i] else np.full(3, np.nan) for i in range(len(horizons))])
        epr_l.append([info['n'][i]['episode']['r'] if done[i] else np.nan for i in range(len(horizons))])
        epl_l.append([info['n'][i]['episode']['l'] if done[i] else np.nan for i in range(len(horizons))])
        viol_l.append([info['n'][i]['episode']['constraint_violation'] if done[i] else np.nan for i in range(len(horizons))])
    put('vec_env/actions', actions); put('vec_env/obs', obs_l); put('vec_env/rew', rew_l); put('vec_env/done', done_l)
    put('vec_env/terminal_observation', term_l); put('vec_env/episode_r', epr_l); put('vec_env/episode_l', epl_l)
    put('vec_env/episode_constraint_violation', viol_l)
    put('vec_env/return_queue', list(venv.return_queue)); put('vec_env/length_queue', list(venv.length_queue))
    put('vec_env/violation_queue', list(venv.queued_stats['constraint_violation']))
    put('vec_env/running_return', venv.episode_return); put('vec_env/running_length', venv.episode_length)
    MANIFEST['groups']['vec_env'] = dict(horizons=horizons, steps=T,
                                         note='VecRecordEpisodeStatistics(DummyVecEnv([ToyEnv(i, horizon_i)])) + a queued constraint_violation '
                                              'tracker; NaN where the key is absent (env not done at that step)')


def gen_normalization(rng):
    rms = NS['RunningMeanStd'](shape=(5,))
    batches, means, variances, counts = [], [], [], []
    for k, n in enumerate([7, 1, 64, 1000, 3]):
        arr = rng.normal(loc=rng.normal(size=5) * 3, scale=rng.uniform(0.1, 4.0, size=5), size=(n, 5))
        rms.update(arr)
        put('normalization/rms/batch%d' % k, arr)
        means.append(rms.mean.copy()); variances.append(rms.var.copy()); counts.append(rms.count)
    put('normalization/rms/mean', means); put('normalization/rms/var', variances); put('normalization/rms/count', counts)
    # MeanStdNormalizer.__call__ (:125-133): update + normalise + clip, then read_only
    msn = NS['MeanStdNormalizer'](shape=(4,), clip=2.5)
    outs = []
    for k in range(4):
        x = rng.normal(size=(50, 4)) * [1.0, 10.0, 0.1, 3.0] + [0.0, 5.0, -2.0, 1.0]
        if k == 3:
            msn.set_read_only()
        put('normalization/msn/x%d' % k, x)
        put('normalization/msn/y%d' % k, msn(x.copy()))
    put('normalization/msn/mean', msn.rms.mean); put('normalization/msn/var', msn.rms.var)
    # RewardStdNormalizer.__call__ (:172-189) incl. its literal  self.ret[dones.astype(np.long)] = 0
    rsn = NS['RewardStdNormalizer'](gamma=0.95, clip=5.0)
    for k in range(6):
        r = rng.normal(size=12) * 2.0
        d = rng.uniform(size=12) < 0.25
        put('normalization/rsn/r%d' % k, r); put('normalization/rsn/d%d' % k, d)
        put('normalization/rsn/out%d' % k, rsn(r.copy(), d.copy()))
        put('normalization/rsn/ret%d' % k, rsn.ret.copy())
    put('normalization/rsn/var', [rsn.rms.var])
    MANIFEST['groups']['normalization'] = dict(note='rms: RunningMeanStd.update over 5 batches (state after each); msn: '
                                                    'MeanStdNormalizer(shape=(4,), clip=2.5), 4 calls, read_only on the last; rsn: '
                                                    'RewardStdNormalizer(gamma=0.95, clip=5.0), 6 calls (legacy index reset semantics)')


def gen_defaults():
    """The default tables of the reference's Quadrotor class (class attributes and one __init__ statement), executed; stored as JSON."""
    h = Holder()
    types.MethodType(NS['init_state_labels'], h)()
    MANIFEST['groups']['defaults'] = dict(
        INIT_STATE_LABELS={str(int(k)): v for k, v in h.INIT_STATE_LABELS.items()},
        INIT_STATE_RAND_INFO=NS['Quadrotor_INIT_STATE_RAND_INFO'], INERTIAL_PROP_RAND_INFO=NS['Quadrotor_INERTIAL_PROP_RAND_INFO'],
        TASK_INFO=NS['Quadrotor_TASK_INFO'], DISTURBANCE_MODES=NS['Quadrotor_DISTURBANCE_MODES'],
        note='Quadrotor.INIT_STATE_RAND_INFO / INERTIAL_PROP_RAND_INFO / TASK_INFO / DISTURBANCE_MODES (class attributes, quadrotor.py:31-146) and '
             'self.INIT_STATE_LABELS (quadrotor.py:207-212), as the executed statements define them')


def gen_rk4():
    """rk_discrete's Runge-Kutta statements (mpc_utils.py:93-97), executed on NumPy vectors with the ORACLE's continuous model as `f`:
    what is pinned is the integration rule (stages, weights, zero-order-hold input), for every system and two step sizes."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import ref_model as rm
    rng = np.random.default_rng(20261006)
    cases = []
    for system in (1, 2, 3, 'cartpole'):
        p = rm.CartPoleParams() if system == 'cartpole' else rm.QuadParams()
        f = rm.F_BY_SYSTEM[system]
        nx, nu = rm.NX[system], rm.NU[system]
        for dt in (1.0 / 60.0, 1.0 / 50.0, 1.0 / 240.0):
            xs = rng.normal(size=(8, nx)) * 0.4
            us = rng.uniform(-3, 3, size=(8, nu)) if system == 'cartpole' else rng.uniform(0.03, 0.14, size=(8, nu))
            out = []
            for x, u in zip(xs, us):
                NS['X'], NS['U'], NS['dt'] = x.copy(), u.copy(), dt
                NS['f'] = lambda X, U: np.asarray(f(list(X), list(U), p), dtype=np.float64)
                out.append(NS['rk4_statements']())
            key = 'rk4/%s/%d' % (system, len(cases))
            put(key + '/x', xs); put(key + '/u', us); put(key + '/x_next', out)
            cases.append(dict(key=key, system=system, dt=dt))
    for k in ('X', 'U', 'dt', 'f'):
        NS.pop(k, None)
    MANIFEST['groups']['rk4'] = dict(cases=cases, note='x_next = the k1 .. k4 / x_next statements of rk_discrete (mpc_utils.py:93-97) with f = '
                                                       'oracle.ref_model.F_BY_SYSTEM[system] (nominal parameters), X / U NumPy vectors')


def gen_quadratic_cost():
    """cost_func = 0.5 (X - Xr)' Q (X - Xr) + 0.5 (U - Ur)' R (U - Ur) (quadrotor.py:682), executed on NumPy columns: identity, diagonal and
    full (non-symmetric) weights for every system size."""
    rng = np.random.default_rng(20261007)
    cases = []
    fn = types.MethodType(NS['quadratic_cost_statement'], Holder())
    for nx, nu in ((2, 1), (6, 2), (12, 4), (4, 1)):
        for kind in ('identity', 'diagonal', 'full'):
            Q = np.eye(nx) if kind == 'identity' else (np.diag(rng.uniform(0.1, 5, nx)) if kind == 'diagonal' else rng.normal(size=(nx, nx)))
            R = np.eye(nu) if kind == 'identity' else (np.diag(rng.uniform(0.1, 5, nu)) if kind == 'diagonal' else rng.normal(size=(nu, nu)))
            xs, xr, us, ur = rng.normal(size=(8, nx)), rng.normal(size=(8, nx)), rng.normal(size=(8, nu)), rng.normal(size=(8, nu))
            out = []
            for k in range(8):
                NS.update(X=xs[k].reshape(nx, 1), Xr=xr[k].reshape(nx, 1), U=us[k].reshape(nu, 1), Ur=ur[k].reshape(nu, 1), Q=Q, R=R)
                out.append(float(np.asarray(fn()).reshape(())))
            key = 'quadratic_cost/%d' % len(cases)
            for name, v in (('Q', Q), ('R', R), ('x', xs), ('xr', xr), ('u', us), ('ur', ur), ('l', out)):
                put(key + '/' + name, v)
            cases.append(dict(key=key, nx=nx, nu=nu, weights=kind))
    for k in ('X', 'Xr', 'U', 'Ur', 'Q', 'R'):
        NS.pop(k, None)
    MANIFEST['groups']['quadratic_cost'] = dict(cases=cases, note='l = the cost_func statement of Quadrotor._setup_symbolic (quadrotor.py:682) on NumPy '
                                                                  'columns; the reward of cost: quadratic is -l (quadrotor.py:918-930)')


def gen_vehicle():
    """The vehicle's constants and its explicit NumPy rigid-body statements, executed."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(20261005)
    h = Holder(GRAVITY_ACC=9.8, DRONE_MODEL=NS['DroneModel'].CF2X)                       # base_aviary.py:90, :64
    bind(h, NS, ['_parse_urdf_parameters'])
    names = ('MASS', 'L', 'THRUST2WEIGHT_RATIO', 'J', 'J_INV', 'KF', 'KM', 'COLLISION_H', 'COLLISION_R', 'COLLISION_Z_OFFSET',
             'MAX_SPEED_KMH', 'GND_EFF_COEFF', 'PROP_RADIUS', 'DRAG_COEFF', 'DW_COEFF_1', 'DW_COEFF_2', 'DW_COEFF_3', 'PWM2RPM_SCALE',
             'PWM2RPM_CONST', 'MIN_PWM', 'MAX_PWM')                                        # the order of base_aviary.py:100-120
    values = h._parse_urdf_parameters(os.path.join(REF, 'envs/gym_pybullet_drones/assets/cf2x.urdf'))
    for k, v in zip(names, values):
        setattr(h, k, v)
        put('vehicle/urdf/' + k, v)
    types.MethodType(NS['vehicle_constants'], h)()
    for k in ('GRAVITY', 'HOVER_RPM', 'MAX_RPM', 'MAX_THRUST', 'MAX_XY_TORQUE', 'MAX_Z_TORQUE', 'GND_EFF_H_CLIP'):
        put('vehicle/const/' + k, getattr(h, k))
    # explicit dynamics: random motor speeds around hover, attitudes up to +-1.2 rad, body rates up to +-8 rad/s
    n = 64
    rpm = h.HOVER_RPM * rng.uniform(0.5, 1.45, size=(n, 4))
    rpy = rng.uniform(-1.2, 1.2, size=(n, 3))
    rates = rng.uniform(-8.0, 8.0, size=(n, 3))
    rot = Rotation.from_euler('xyz', rpy).as_matrix()          # extrinsic x-y-z = Rz(yaw) Ry(pitch) Rx(roll): what PyBullet's
    acc, rdot = [], []                                          # getMatrixFromQuaternion(getQuaternionFromEuler(rpy)) returns
    fn = types.MethodType(NS['explicit_dynamics'], h)
    for i in range(n):
        NS['rpm'], NS['rotation'], NS['rpy_rates'] = rpm[i], rot[i], rates[i]
        a, r = fn()
        acc.append(a); rdot.append(r)
    put('vehicle/dynamics/rpm', rpm); put('vehicle/dynamics/rpy', rpy); put('vehicle/dynamics/rates', rates)
    put('vehicle/dynamics/rotation', rot); put('vehicle/dynamics/acc', acc); put('vehicle/dynamics/rates_deriv', rdot)
    fn = types.MethodType(NS['pyb_forces_and_yaw_torque'], h)
    forces, ztq = [], []
    for i in range(n):
        NS['rpm'] = rpm[i]
        f, z = fn()
        forces.append(f); ztq.append(z)
    put('vehicle/physics/forces', forces); put('vehicle/physics/z_torque', ztq)
    # ground effect: per-propeller forces for propeller heights from below the clip to 1 m
    heights = np.concatenate([rng.uniform(0.0, 0.05, size=(16, 4)), rng.uniform(0.05, 1.0, size=(16, 4))])
    gnd = []
    fn = types.MethodType(NS['ground_effect_forces'], h)
    for i in range(heights.shape[0]):
        links = np.empty((5, 1), dtype=object)                 # getLinkStates(...)[i][0] = world position of link i (4 propellers + CoM)
        for q in range(4):
            links[q, 0] = (0.0, 0.0, float(heights[i, q]))
        links[4, 0] = (0.0, 0.0, float(heights[i].mean()))
        NS['rpm'], NS['link_states'] = rpm[i], links
        gnd.append(fn())
    put('vehicle/gnd/rpm', rpm[:heights.shape[0]]); put('vehicle/gnd/prop_heights', heights); put('vehicle/gnd/forces', gnd)
    # drag: link-frame force for random velocities
    vel = rng.uniform(-3.0, 3.0, size=(n, 3))
    drag = []
    fn = types.MethodType(NS['drag_force'], h)
    NS['nth_drone'] = 0
    for i in range(n):
        NS['rpm'], NS['base_rot'] = rpm[i], rot[i]
        h.vel = vel[i][None, :]
        drag.append(fn())
    put('vehicle/drag/vel', vel); put('vehicle/drag/link_frame_force', drag)
    for k in ('rpm', 'rotation', 'rpy_rates', 'link_states', 'base_rot', 'nth_drone'):
        NS.pop(k, None)
    MANIFEST['groups']['vehicle'] = dict(note='urdf/*: _parse_urdf_parameters(assets/cf2x.urdf) in the order of base_aviary.py:100-120; const/*: '
                                              'the "Compute constants" statements (:138-147); dynamics/*: statements :540-558 of _dynamics for 64 '
                                              'samples -- `rotation` (PyBullet\'s getMatrixFromQuaternion in the reference) is an INPUT, computed '
                                              'by scipy.spatial.transform.Rotation.from_euler("xyz", rpy); acc = no_pybullet_dyn_accs, rates_deriv = '
                                              'rpy_rates_deriv; physics/*: the forces / z_torque statements of _physics (:420-423) for the same rpm; gnd/*: the prop_heights / clip / gnd_effects statements of _ground_effect (:455-461) for given '
                                              'propeller positions (`link_states`, PyBullet\'s getLinkStates in the reference, is an INPUT); drag/*: the two NumPy statements of _drag (:489-491), the force PyBullet then '
                                              'applies in the LINK frame')


def main():
    rng = np.random.default_rng(20241004)
    gen_motor_map(rng)
    gen_trajectory()
    gen_transform(rng)
    gen_x_goal()
    gen_constraints(rng)
    gen_step_glue(rng)
    gen_step_glue_device(np.random.default_rng(20261004))      # its own stream: the earlier groups keep their recorded values
    gen_gae(rng)
    gen_normalization(rng)
    gen_discretize(rng)        # (appended: earlier groups keep the random stream they were first written with)
    gen_vec_env(rng)
    gen_spaces(rng)            # (appended; draws nothing)
    gen_randomization(rng)     # (appended; its own RandomState)
    gen_ppo_buffer(np.random.default_rng(77))     # (appended; its own generator)
    gen_vehicle()              # (appended; its own generator)
    gen_defaults()             # (appended; draws nothing)
    gen_rk4()                  # (appended; its own generator)
    gen_quadratic_cost()       # (appended; its own generator)
    OUT['manifest'] = np.array(json.dumps(MANIFEST))
    np.savez_compressed(os.path.join(HERE, 'reference_numpy.npz'), **OUT)
    print('wrote %d arrays, %d reference symbols executed' % (len(OUT) - 1, len(SOURCES)))
    for f, s, a, b in SOURCES:
        print('  safe_control_gym/%s:%d-%d  %s' % (f, a, b, s))


if __name__ == '__main__':
    main()
