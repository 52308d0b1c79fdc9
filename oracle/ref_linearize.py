"""Oracle: linearisation and quadratic stage cost of the prior model, float64.

Test infrastructure -- see ``oracle/__init__.py``.

Follows /root/reference/safe_control_gym:
  * ``SymbolicModel.setup_linearization`` (math_and_models/symbolic_systems.py:77-92): ``dfdx`` and
    ``dfdu`` are SYMBOLIC Jacobians of ``x_dot`` (CasADi ``cs.jacobian``).  CasADi is absent here, so the
    same operation is done with sympy on the matrix-form dynamics that ``oracle/independent.py`` rebuilds
    from quadrotor.py:628-674 -- symbolic differentiation of the reference's expression, not a hand
    derivation (the HIP kernel holds the hand-derived closed forms; the two meet in the parity tests).
  * ``discretize_linear_system`` (controllers/mpc/mpc_utils.py:24-55; same in lqr/lqr_utils.py:39-60).
  * the quadratic cost and its derivatives (quadrotor.py:678-682, symbolic_systems.py:94-123).
A second, derivation-free check (central differences of ``ref_model.f_*``) lives in
tests/test_oracle_pinning.py.
"""
import numpy as np
import scipy.linalg
import sympy as sp

from oracle import independent

_JAC = {}


def _jac_funcs(system):
    if system not in _JAC:
        X, U, P, Xdot = independent.symbolic(system)
        _JAC[system] = (sp.lambdify([X, U, P], Xdot.jacobian(X), 'numpy'),
                        sp.lambdify([X, U, P], Xdot.jacobian(U), 'numpy'))
    return _JAC[system]


def df(system, x, u, p):
    """(dfdx (nx, nx), dfdu (nx, nu)) at one (x, u); ``system`` in {1, 2, 3, 'cartpole'}."""
    fa, fb = _jac_funcs(system)
    pv = independent.params_vector(system, p)
    x = [float(v) for v in x]
    u = [float(v) for v in u]
    return np.asarray(fa(x, u, pv), dtype=np.float64), np.asarray(fb(x, u, pv), dtype=np.float64)


def discretize_linear_system(A, B, dt, exact=False):
    """mpc_utils.py:24-55."""
    nx, nu = A.shape[1], B.shape[1]
    if exact:
        M = np.zeros((nx + nu, nx + nu))
        M[:nx, :nx] = A
        M[:nx, nx:] = B
        Md = scipy.linalg.expm(M * dt)
        return Md[:nx, :nx], Md[:nx, nx:]
    return np.eye(nx) + A * dt, B * dt


def quadratic_loss(x, u, xr, ur, Q, R):
    """l, l_x, l_xx, l_u, l_uu, l_xu of  1/2 ex'Q ex + 1/2 eu'R eu  (quadrotor.py:682; the derivatives are what
    cs.jacobian yields for a general, not necessarily symmetric, Q and R: symbolic_systems.py:108-113)."""
    ex = np.asarray(x, dtype=np.float64) - np.asarray(xr, dtype=np.float64)
    eu = np.asarray(u, dtype=np.float64) - np.asarray(ur, dtype=np.float64)
    Q = np.asarray(Q, dtype=np.float64)
    R = np.asarray(R, dtype=np.float64)
    return {'l': 0.5 * ex @ Q @ ex + 0.5 * eu @ R @ eu,
            'l_x': 0.5 * (Q + Q.T) @ ex, 'l_xx': 0.5 * (Q + Q.T),
            'l_u': 0.5 * (R + R.T) @ eu, 'l_uu': 0.5 * (R + R.T),
            'l_xu': np.zeros((ex.shape[0], eu.shape[0]))}


def _f(system, x, u, p):
    from oracle import ref_model as rm
    return np.asarray(rm.F_BY_SYSTEM[system](list(x), list(u), p), dtype=np.float64)


def fc_linear(system, x_eval, u_eval, x, u, p):
    """SymbolicModel.fc_linear_func (math_and_models/symbolic_systems.py:93-101):
    x_dot_linear = x_dot + dfdx (x_eval - x) + dfdu (u_eval - u), Jacobians by symbolic differentiation (``df``)."""
    A, B = df(system, x, u, p)
    return _f(system, x, u, p) + A @ (np.asarray(x_eval, dtype=np.float64) - np.asarray(x, dtype=np.float64)) \
        + B @ (np.asarray(u_eval, dtype=np.float64) - np.asarray(u, dtype=np.float64))


def fd_linear(system, x_eval, u_eval, x, u, p, dt, method='expm'):
    """SymbolicModel.fd_linear_func (:102-107): the linearised ODE  z' = fc_linear(z, u_eval; x, u)  integrated over
    [0, dt] from z(0) = x_eval.  The reference hands it to CVODES; the ODE is affine (z' = A z + c), so its flow is
    ``expm([[A, c], [0, 0]] dt)`` -- ``method='ivp'`` integrates it numerically instead (DOP853, rtol 1e-12), the CVODES
    stand-in the tests compare the closed form with."""
    A, _ = df(system, x, u, p)
    c = fc_linear(system, np.zeros(len(x)), u_eval, x, u, p)
    z0 = np.asarray(x_eval, dtype=np.float64)
    nx = len(z0)
    if method == 'ivp':
        from scipy.integrate import solve_ivp
        sol = solve_ivp(lambda t, z: A @ z + c, (0.0, dt), z0, method='DOP853', rtol=1e-12, atol=1e-14)
        return sol.y[:, -1]
    M = np.zeros((nx + 1, nx + 1))
    M[:nx, :nx], M[:nx, nx] = A, c
    E = scipy.linalg.expm(M * dt)
    return E[:nx, :nx] @ z0 + E[:nx, nx]
