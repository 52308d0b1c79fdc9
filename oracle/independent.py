"""Oracle cross-check: an independent derivation of the prior dynamics (sympy, matrix form) and a
high-accuracy integration (scipy DOP853) standing in for the reference's CVODES ``fd_func``.

Test infrastructure -- see ``oracle/__init__.py``.

This deliberately does NOT share code with ``ref_model.py``: it rebuilds R = Rz Ry Rx, the skew
matrix and the Euler-rate matrix symbolically from the matrix expressions written in
/root/reference/safe_control_gym/envs/gym_pybullet_drones/quadrotor.py:628-674 and
math_and_models/transformations.py:125-182, then lambdifies them.  ``ref_model.f_quad3d`` is the
hand-expanded scalar form; the two must agree to rounding (tests/test_oracle_pinning.py).
"""
import numpy as np
import sympy as sp
from scipy.integrate import solve_ivp


def _sym_quad3d():
    x, xd, y, yd, z, zd, phi, th, psi, p, q, r = sp.symbols('x xd y yd z zd phi th psi p q r')
    f1, f2, f3, f4 = sp.symbols('f1 f2 f3 f4')
    m, g, l, Ixx, Iyy, Izz, gamma = sp.symbols('m g l Ixx Iyy Izz gamma')
    Rz = sp.Matrix([[sp.cos(psi), -sp.sin(psi), 0], [sp.sin(psi), sp.cos(psi), 0], [0, 0, 1]])
    Ry = sp.Matrix([[sp.cos(th), 0, sp.sin(th)], [0, 1, 0], [-sp.sin(th), 0, sp.cos(th)]])
    Rx = sp.Matrix([[1, 0, 0], [0, sp.cos(phi), -sp.sin(phi)], [0, sp.sin(phi), sp.cos(phi)]])
    Rob = Rz * Ry * Rx
    J = sp.diag(Ixx, Iyy, Izz)
    Jinv = sp.diag(1 / Ixx, 1 / Iyy, 1 / Izz)
    w = sp.Matrix([p, q, r])
    skew = sp.Matrix([[0, -r, q], [r, 0, -p], [-q, p, 0]])
    acc = Rob * sp.Matrix([0, 0, f1 + f2 + f3 + f4]) / m - sp.Matrix([0, 0, g])
    Mb = sp.Matrix([l / sp.sqrt(2) * (f1 + f2 - f3 - f4),
                    l / sp.sqrt(2) * (-f1 + f2 + f3 - f4),
                    gamma * (f1 - f2 + f3 - f4)])
    rate_dot = Jinv * (Mb - skew * J * w)
    E = sp.Matrix([[1, sp.sin(phi) * sp.tan(th), sp.cos(phi) * sp.tan(th)],
                   [0, sp.cos(phi), -sp.sin(phi)],
                   [0, sp.sin(phi) / sp.cos(th), sp.cos(phi) / sp.cos(th)]])
    ang_dot = E * w
    X = [x, xd, y, yd, z, zd, phi, th, psi, p, q, r]
    U = [f1, f2, f3, f4]
    P = [m, g, l, Ixx, Iyy, Izz, gamma]
    Xdot = sp.Matrix([xd, acc[0], yd, acc[1], zd, acc[2], ang_dot[0], ang_dot[1], ang_dot[2],
                      rate_dot[0], rate_dot[1], rate_dot[2]])
    return X, U, P, Xdot


def _sym_quad2d():
    x, xd, z, zd, th, thd, T1, T2 = sp.symbols('x xd z zd th thd T1 T2')
    m, g, l, Iyy = sp.symbols('m g l Iyy')
    Xdot = sp.Matrix([xd, sp.sin(th) * (T1 + T2) / m, zd, sp.cos(th) * (T1 + T2) / m - g, thd,
                      l * (T2 - T1) / Iyy / sp.sqrt(2)])
    return [x, xd, z, zd, th, thd], [T1, T2], [m, g, l, Iyy], Xdot


def _sym_cartpole():
    """Lagrangian derivation (independent of the closed form in ref_model.f_cartpole): uniform
    pole of half-length l and mass m hinged on a cart of mass M, theta from upright."""
    t = sp.symbols('t')
    m, M, l, g, F = sp.symbols('m M l g F')
    xc = sp.Function('x')(t)
    th = sp.Function('th')(t)
    # pole centre of mass and energies (uniform rod of length 2l: I_cm = m (2l)^2 / 12 = m l^2 / 3)
    px = xc + l * sp.sin(th)
    pz = l * sp.cos(th)
    T = sp.Rational(1, 2) * M * xc.diff(t) ** 2 + sp.Rational(1, 2) * m * (px.diff(t) ** 2 + pz.diff(t) ** 2) \
        + sp.Rational(1, 2) * (m * l ** 2 / 3) * th.diff(t) ** 2
    V = m * g * pz
    L = T - V
    eq1 = sp.diff(L.diff(xc.diff(t)), t) - L.diff(xc) - F
    eq2 = sp.diff(L.diff(th.diff(t)), t) - L.diff(th)
    xdd, thdd = sp.symbols('xdd thdd')
    subs = {xc.diff(t, 2): xdd, th.diff(t, 2): thdd}
    sol = sp.solve([eq1.subs(subs), eq2.subs(subs)], [xdd, thdd], dict=True)[0]
    X = sp.symbols('x xd th thd')
    rep = {xc.diff(t): X[1], th.diff(t): X[3], xc: X[0], th: X[2]}
    Xdot = sp.Matrix([X[1], sp.simplify(sol[xdd].subs(rep)), X[3], sp.simplify(sol[thdd].subs(rep))])
    return list(X), [F], [l, m, M, g], Xdot


def _sym_quad1d():
    z, zd, T, m, g = sp.symbols('z zd T m g')
    return [z, zd], [T], [m, g], sp.Matrix([zd, T / m - g])


_SYM = {}
_CACHE = {}


def symbolic(system):
    """(X, U, P, Xdot) sympy objects of one system (cached)."""
    if system not in _SYM:
        _SYM[system] = {3: _sym_quad3d, 2: _sym_quad2d, 1: _sym_quad1d, 'cartpole': _sym_cartpole}[system]()
    return _SYM[system]


def f_independent(system):
    if system not in _CACHE:
        X, U, P, Xdot = symbolic(system)
        _CACHE[system] = sp.lambdify([X, U, P], Xdot, 'numpy')
    return _CACHE[system]


def params_vector(system, p):
    if system == 3:
        return [p.m, p.g, p.l, p.ixx, p.iyy, p.izz, p.km / p.kf]
    if system == 2:
        return [p.m, p.g, p.l, p.iyy]
    if system == 1:
        return [p.m, p.g]
    return [p.l, p.m, p.M, p.g]


def exact_step(system, x, u, p, dt):
    """x(t+dt) with ZOH input by DOP853 at rtol 1e-13: the CVODES / 'exact' stand-in."""
    f = f_independent(system)
    pv = params_vector(system, p)
    sol = solve_ivp(lambda t, y: np.asarray(f(y, u, pv), dtype=float).ravel(), (0.0, dt),
                    np.asarray(x, dtype=float), method='DOP853', rtol=1e-13, atol=1e-15)
    return sol.y[:, -1]


_JAC = {}


def _jacobians(system):
    if system not in _JAC:
        X, U, P, Xdot = symbolic(system)
        _JAC[system] = (sp.lambdify([X, U, P], Xdot.jacobian(X), 'numpy'), sp.lambdify([X, U, P], Xdot.jacobian(U), 'numpy'))
    return _JAC[system]


def exact_step_sensitivities(system, x, u, p, dt):
    """(x(dt), d x(dt) / d x0, d x(dt) / d u) with ZOH input: DOP853 at rtol 1e-13 on the variational equations
    Sx' = A(x) Sx, Sx(0) = I;  Su' = A(x) Su + B(x), Su(0) = 0 -- what CVODES' forward sensitivities of the reference's
    ``fd_func`` (symbolic_systems.py:70-73) converge to."""
    f = f_independent(system)
    fa, fb = _jacobians(system)
    pv = params_vector(system, p)
    x = np.asarray(x, dtype=float)
    nx, nu = x.size, np.asarray(u).size

    def rhs(t, z):
        y = z[:nx]
        A = np.asarray(fa(y, u, pv), dtype=float).reshape(nx, nx)
        B = np.asarray(fb(y, u, pv), dtype=float).reshape(nx, nu)
        Sx, Su = z[nx:nx + nx * nx].reshape(nx, nx), z[nx + nx * nx:].reshape(nx, nu)
        return np.concatenate([np.asarray(f(y, u, pv), dtype=float).ravel(), (A @ Sx).ravel(), (A @ Su + B).ravel()])

    z0 = np.concatenate([x, np.eye(nx).ravel(), np.zeros(nx * nu)])
    z = solve_ivp(rhs, (0.0, dt), z0, method='DOP853', rtol=1e-13, atol=1e-15).y[:, -1]
    return z[:nx], z[nx:nx + nx * nx].reshape(nx, nx), z[nx + nx * nx:].reshape(nx, nu)
