"""Oracle: Constraint / ConstraintList evaluation (float64, per environment).

Test infrastructure -- see ``oracle/__init__.py``.

Reference anchors (relative to /root/reference/safe_control_gym/envs/constraints.py):
  * :35-84    Constraint.__init__   (dim by constrained_variable, filter F = I[active_dims])
  * :106-119  get_value             round_8( sym_func(v) ),  v = state | RAW action (:160-172)
  * :121-158  is_violated (strict -> >=0 else >0), is_almost_active (c + tol > 0)
  * :179-223  QuadraticContstraint  v' F' P F v - b   (scalar)
  * :226-270  LinearConstraint      A F v - b
  * :273-304  BoundedConstraint     A = [-I; I], b = [-lb; ub]
  * :307-368  DefaultConstraint     bounds of env.state_space / env.action_space, cast f64
  * :371-415  SymmetricStateConstraint  |F x| - bound   (cartpole + rl_reward only)
  * :426-583  ConstraintList        concatenation in spec order, split by cumulative lengths
  * :594-612  create_constraint_list (yaml factory)

Deviation noted: for ``input_and_state`` the reference builds ``np.array((state, action))``
from two different-length arrays (:119,:170) which is not a numeric vector; the oracle (and
the engine) define v = concat(state, raw action), consistent with ``dim = nx + nu`` (:63).
"""
import numpy as np


class RefConstraint:
    rounding = 8

    def __init__(self, env, constrained_variable, strict=False, active_dims=None, tolerance=None, **kw):
        self.constrained_variable = str(constrained_variable)
        if self.constrained_variable == 'state':
            self.dim = env.state_dim
        elif self.constrained_variable == 'input':
            self.dim = env.action_dim
        elif self.constrained_variable == 'input_and_state':
            self.dim = env.state_dim + env.action_dim
        else:
            raise NotImplementedError('[ERROR] invalid constrained_variable (use STATE, INPUT or INPUT_AND_STATE).')
        self.strict = strict
        if active_dims is not None:
            if isinstance(active_dims, int):
                active_dims = [active_dims]
            assert isinstance(active_dims, (list, np.ndarray))
            assert len(active_dims) <= self.dim
            assert all(isinstance(n, int) for n in active_dims)
            assert all(n < self.dim for n in active_dims)
            assert len(active_dims) == len(set(active_dims))
            self.constraint_filter = np.eye(self.dim)[active_dims]
            self.dim = len(active_dims)
        else:
            self.constraint_filter = np.eye(self.dim)
        self.tolerance = None if tolerance is None else np.array(tolerance, ndmin=1)

    def _var(self, env):
        if self.constrained_variable == 'state':
            return np.asarray(env.state, dtype=float)
        if self.constrained_variable == 'input':
            return np.asarray(env.current_raw_input_action, dtype=float)
        return np.concatenate([np.asarray(env.state, dtype=float),
                               np.asarray(env.current_raw_input_action, dtype=float)])

    def get_value(self, env):
        v = np.array(self._var(env), ndmin=1)
        return np.round(np.atleast_1d(np.squeeze(self.sym_func(v))), decimals=self.rounding)

    def is_violated(self, env, c_value=None):
        if c_value is None:
            c_value = self.get_value(env)
        if self.strict:
            return bool(np.any(np.greater_equal(c_value, 0.)))
        return bool(np.any(np.greater(c_value, 0.)))

    def is_almost_active(self, env, c_value=None):
        if self.tolerance is None:
            return False
        if c_value is None:
            c_value = self.get_value(env)
        return bool(np.any(np.greater(c_value + self.tolerance, 0.)))

    def _check_tol(self):
        if self.tolerance is not None and len(self.tolerance) != self.num_constraints:
            raise ValueError('[ERROR] the tolerance dimension does not match the number of constraints.')


class RefQuadratic(RefConstraint):
    def __init__(self, env, P, b, constrained_variable, strict=False, active_dims=None, tolerance=None):
        super().__init__(env, constrained_variable, strict=strict, active_dims=active_dims, tolerance=tolerance)
        P = np.array(P, ndmin=1)
        assert P.shape == (self.dim, self.dim)
        assert isinstance(b, float)
        self.P, self.b = P, b
        self.num_constraints = 1
        F = self.constraint_filter
        self.sym_func = lambda x: x.T @ F.T @ self.P @ F @ x - self.b
        self._check_tol()


class RefLinear(RefConstraint):
    def __init__(self, env, A, b, constrained_variable, strict=False, active_dims=None, tolerance=None):
        super().__init__(env, constrained_variable, strict=strict, active_dims=active_dims, tolerance=tolerance)
        A = np.array(A, ndmin=1)
        b = np.array(b, ndmin=1)
        assert A.shape[1] == self.dim
        assert b.shape[0] == A.shape[0]
        self.A, self.b = A, b
        self.num_constraints = A.shape[0]
        self.sym_func = lambda x: self.A @ self.constraint_filter @ x - self.b
        self._check_tol()


class RefBounded(RefLinear):
    def __init__(self, env, lower_bounds, upper_bounds, constrained_variable, strict=False,
                 active_dims=None, tolerance=None):
        self.lower_bounds = np.array(lower_bounds, ndmin=1)
        self.upper_bounds = np.array(upper_bounds, ndmin=1)
        dim = self.lower_bounds.shape[0]
        A = np.vstack((-np.eye(dim), np.eye(dim)))
        b = np.hstack((-self.lower_bounds, self.upper_bounds))
        super().__init__(env, A, b, constrained_variable, strict=strict, active_dims=active_dims,
                         tolerance=tolerance)


class RefDefault(RefBounded):
    def __init__(self, env, constrained_variable, lower_bounds=None, upper_bounds=None, strict=False,
                 tolerance=None):
        if constrained_variable == 'state':
            low, high = env.state_space_low, env.state_space_high
        elif constrained_variable == 'input':
            low, high = env.action_space_low, env.action_space_high
        else:
            raise NotImplementedError('[ERROR] DefaultConstraint can only be of type STATE or INPUT')
        if upper_bounds is None:
            upper_bounds = high
        else:
            upper_bounds = np.array(upper_bounds, ndmin=1)
            assert len(upper_bounds) == len(high)
        if lower_bounds is None:
            lower_bounds = low
        else:
            lower_bounds = np.array(lower_bounds, ndmin=1)
            assert len(lower_bounds) == len(low)
        super().__init__(env, lower_bounds=np.asarray(lower_bounds).astype(np.float64),
                         upper_bounds=np.asarray(upper_bounds).astype(np.float64),
                         constrained_variable=constrained_variable, strict=strict,
                         active_dims=None, tolerance=tolerance)


class RefSymmetricState(RefBounded):
    def __init__(self, env, constrained_variable, bound, strict=False, active_dims=None, tolerance=None, **kw):
        assert bound is not None
        self.bound = np.array(bound, ndmin=1)
        self._sym_tol_len = len(self.bound)
        super().__init__(env, lower_bounds=-self.bound, upper_bounds=self.bound,
                         constrained_variable=constrained_variable, strict=strict,
                         active_dims=active_dims, tolerance=tolerance)
        assert env.NAME == 'cartpole'
        assert env.COST == 'rl_reward'
        self.num_constraints = self.bound.shape[0]

    def get_value(self, env):
        return np.abs(self.constraint_filter @ np.asarray(env.state, dtype=float)) - self.bound

    def _check_tol(self):
        if self.tolerance is not None and len(self.tolerance) != len(self.bound):
            raise ValueError('[ERROR] the tolerance dimension does not match the number of constraints.')


REF_CONSTRAINTS = {
    'linear_constraint': RefLinear,
    'quadratic_constraint': RefQuadratic,
    'bounded_constraint': RefBounded,
    'default_constraint': RefDefault,
    'symmetric_state': RefSymmetricState,              # cartpole-only in upstream
    'symmetric_state_constraint': RefSymmetricState,
}


class RefConstraintList:
    def __init__(self, constraints):
        self.constraints = constraints
        self.constraint_lengths = [c.num_constraints for c in constraints]
        self.constraint_indices = np.cumsum(self.constraint_lengths[:-1])
        self.num_constraints = sum(self.constraint_lengths)
        self.state_constraints = [c for c in constraints if c.constrained_variable == 'state']

    def get_values(self, env, only_state=False):
        cons = self.state_constraints if only_state else self.constraints
        return np.concatenate([c.get_value(env) for c in cons])

    def is_violated(self, env, c_value=None):
        if c_value is not None:
            splits = np.split(c_value, self.constraint_indices)
            return any(c.is_violated(env, c_value=s) for c, s in zip(self.constraints, splits))
        return any(c.is_violated(env) for c in self.constraints)

    def is_almost_active(self, env, c_value=None):
        if c_value is not None:
            splits = np.split(c_value, self.constraint_indices)
            return any(c.is_almost_active(env, c_value=s) for c, s in zip(self.constraints, splits))
        return any(c.is_almost_active(env) for c in self.constraints)


def create_ref_constraint_list(specs, env):
    out = []
    for spec in specs:
        assert isinstance(spec, dict)
        assert 'constraint_form' in spec
        form = spec['constraint_form']
        assert form in REF_CONSTRAINTS
        cfg = {k: v for k, v in spec.items() if k != 'constraint_form'}
        out.append(REF_CONSTRAINTS[form](env, **cfg))
    return RefConstraintList(out)
