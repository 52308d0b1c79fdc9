"""Prior ("symbolic") model handed to controllers through ``env.symbolic`` -- host side, batched NumPy.

Mirrors the public surface of the reference's CasADi ``SymbolicModel``
(math_and_models/symbolic_systems.py:8-123) without CasADi: ``nx, nu, ny, dt``,
``fc_func`` (:68), ``fd_func`` (:70-73), ``g_func`` (:75), ``df_func`` (:84-86), ``dg_func``
(:89-91), ``fc_linear_func`` / ``fd_linear_func`` / ``g_linear_func`` (:93-112), ``loss`` (:119-123).  Every function takes keyword arrays like the CasADi ``Function``
objects do and returns a dict with the same output names; inputs may carry leading batch
dimensions (``x`` of shape (..., nx)).

Differences, all deliberate:
  * ``fd_func`` is the flow over ``[0, dt]`` to CasADi's default CVODES tolerances (rtol 1e-6, atol 1e-8) like the
    reference's, but integrated by RK4 with step-doubling error control and order-5 extrapolation in float64
    (SUNDIALS is not available): NumPy here, ``scg_model_flow`` for device tensors.  ``fd_jac_func``
    gives the integrator's forward sensitivities (``jac_xf_x0`` / ``jac_xf_p``).  ``rk4_func`` is ONE fixed RK4 step
    (the reference's ``rk_discrete``, controllers/mpc/mpc_utils.py:79-101) -- the discretisation the engine steps with.
  * Jacobians come from a sympy derivation of the same equations, lambdified on first use.

The equations are those of quadrotor.py:594-676 (1-D / 2-D / 3-D) and the textbook cart-pole
(DESIGN.md, row A16).
"""
import numpy as np


# Component-wise closed forms, generic over the math namespace ``m`` (numpy for evaluation, sympy
# for deriving Jacobians): x, u are sequences of components, the result a list of components.
def _quad1d(x, u, c, m):
    return [x[1], u[0] / c['m'] - c['g']]


def _quad2d(x, u, c, m):
    th = x[4]
    T = u[0] + u[1]
    return [x[1], m.sin(th) * T / c['m'], x[3], m.cos(th) * T / c['m'] - c['g'],
            x[5], c['l'] * (u[1] - u[0]) / c['Iyy'] / m.sqrt(2.0)]


def _quad3d(x, u, c, m):
    phi, th, psi, p, q, r = x[6], x[7], x[8], x[9], x[10], x[11]
    f1, f2, f3, f4 = u
    sph, cph, sth, cth, sps, cps = m.sin(phi), m.cos(phi), m.sin(th), m.cos(th), m.sin(psi), m.cos(psi)
    Tm = (f1 + f2 + f3 + f4) / c['m']
    ax = (cps * sth * cph + sps * sph) * Tm
    ay = (sps * sth * cph - cps * sph) * Tm
    az = cth * cph * Tm - c['g']
    k = c['l'] / m.sqrt(2.0)
    Mx, My, Mz = k * (f1 + f2 - f3 - f4), k * (-f1 + f2 + f3 - f4), c['gamma'] * (f1 - f2 + f3 - f4)
    Ixx, Iyy, Izz = c['Ixx'], c['Iyy'], c['Izz']
    pd = (Mx - (q * Izz * r - r * Iyy * q)) / Ixx
    qd = (My - (r * Ixx * p - p * Izz * r)) / Iyy
    rd = (Mz - (p * Iyy * q - q * Ixx * p)) / Izz
    tth = sth / cth
    return [x[1], ax, x[3], ay, x[5], az,
            p + sph * tth * q + cph * tth * r, cph * q - sph * r, (sph * q + cph * r) / cth,
            pd, qd, rd]


def _cartpole(x, u, c, m):
    th, thd = x[2], x[3]
    Mm, ml = c['m'] + c['M'], c['m'] * c['l']
    s, co = m.sin(th), m.cos(th)
    tmp = (u[0] + ml * thd * thd * s) / Mm
    thdd = (c['g'] * s - co * tmp) / (c['l'] * (4.0 / 3.0 - c['m'] * co * co / Mm))
    return [x[1], tmp - ml * thdd * co / Mm, thd, thdd]


def _is_device_tensor(x):
    return type(x).__module__.startswith('torch') and getattr(x, 'is_cuda', False)


_F = {'quad1d': _quad1d, 'quad2d': _quad2d, 'quad3d': _quad3d, 'cartpole': _cartpole}
_DIMS = {'quad1d': (2, 1), 'quad2d': (6, 2), 'quad3d': (12, 4), 'cartpole': (4, 1)}


class AnalyticModel:
    """Closed-form stand-in for ``SymbolicModel``."""

    def __init__(self, system, consts, dt):
        self.system = system
        self.consts = dict(consts)
        self.dt = dt
        self.nx, self.nu = _DIMS[system]
        self.ny = self.nx
        self.integration_algo = 'rk4 + step doubling'   # the reference's attribute (:30, 'cvodes'): how fd_func integrates
        self._jac = None
        self.device_backend = None        # the env (weak proxy): torch device tensors go to its HIP kernels

    def _backend(self):
        if self.device_backend is None:
            raise RuntimeError('[ERROR] device tensors need the env that owns this model (env.symbolic).')
        return self.device_backend

    # -- dynamics ---------------------------------------------------------------------------
    def _f(self, x, u):
        x = np.asarray(x, dtype=np.float64)
        u = np.asarray(u, dtype=np.float64)
        xs = [x[..., j] for j in range(self.nx)]
        us = [u[..., j] for j in range(self.nu)]
        comps = _F[self.system](xs, us, self.consts, np)
        shape = np.broadcast(*[np.asarray(v) for v in comps]).shape
        return np.stack([np.broadcast_to(np.asarray(v, dtype=np.float64), shape) for v in comps], -1)

    def fc_func(self, x, u):
        if _is_device_tensor(x):
            return {'f': self._backend().model_dynamics(x, u)}
        return {'f': self._f(x, u)}

    def rk4_func(self, x0, p):
        """One RK4 step of length dt with zero-order-hold input p (controllers/mpc/mpc_utils.py:79-101; device tensors: the
        env's own integrator)."""
        if _is_device_tensor(x0):
            return {'xf': self._backend().model_step(x0, p)}
        return {'xf': self._rk4_chain(np.asarray(x0, dtype=np.float64), p, self.dt, 1)}

    def _rk4_chain(self, x, p, dt, m):
        h = dt / m
        for _ in range(m):
            k1 = self._f(x, p)
            k2 = self._f(x + h / 2 * k1, p)
            k3 = self._f(x + h / 2 * k2, p)
            k4 = self._f(x + h * k3, p)
            x = x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        return x

    def _extrapolated(self, x0, p, m):
        y1, y2 = self._rk4_chain(x0, p, self.dt, m), self._rk4_chain(x0, p, self.dt, 2 * m)
        return y2 + (y2 - y1) / 15.0

    def _flow(self, x0, p, rtol, atol, max_doublings=12):
        """Step doubling on RK4, accepted per (x0, p) pair (the scheme of csrc/scg_flow.hip): (extrapolated values, the m of
        each accepted pair)."""
        x0 = np.asarray(x0, dtype=np.float64)
        lead = x0.shape[:-1]
        p = np.broadcast_to(np.asarray(p, dtype=np.float64), lead + (self.nu,)).reshape(-1, self.nu)
        x0 = x0.reshape(-1, self.nx)
        out = np.empty_like(x0)
        ms = np.zeros(x0.shape[0], dtype=np.int64)
        m = 1
        y1 = self._rk4_chain(x0, p, self.dt, m)
        for lvl in range(max_doublings + 1):
            y2 = self._rk4_chain(x0, p, self.dt, 2 * m)
            ok = np.all(np.abs(y2 - y1) <= 15.0 * (atol + rtol * np.abs(y2)), axis=-1)
            take = (ms == 0) & (ok | (lvl == max_doublings))
            out[take] = (y2 + (y2 - y1) / 15.0)[take]
            ms[take] = m
            if np.all(ms > 0):
                break
            y1, m = y2, 2 * m
        return out.reshape(lead + (self.nx,)), ms.reshape(lead)

    def _extrapolated_per_pair(self, x0, p, ms):
        out = np.empty_like(x0)
        for m in np.unique(ms):
            sel = ms == m
            out[sel] = self._extrapolated(x0[sel], p[sel], int(m))
        return out

    def fd_func(self, x0, p, rtol=1e-6, atol=1e-8):
        """The flow of x' = f(x, p) over [0, dt] with constant p, to atol + rtol |x| (symbolic_systems.py:70-73; defaults =
        CasADi's CVODES defaults).  Device tensors: ``scg_model_flow`` (float64 inside; returned in x0's dtype)."""
        if _is_device_tensor(x0):
            return {'xf': self._backend().model_flow(x0, p, dt=self.dt, rtol=rtol, atol=atol).to(x0.dtype)}
        return {'xf': self._flow(x0, p, rtol, atol)[0]}

    def fd_jac_func(self, x0, p, rtol=1e-6, atol=1e-8):
        """Forward sensitivities of ``fd_func`` (the CasADi integrator's ``jac_xf_x0`` / ``jac_xf_p``): central differences of
        the same discretisation (internal numerical differentiation, ~1e-9 relative)."""
        if _is_device_tensor(x0):
            xf, sx, su = self._backend().model_flow(x0, p, dt=self.dt, rtol=rtol, atol=atol, sensitivities=True)
            return {'xf': xf.to(x0.dtype), 'jac_xf_x0': sx.to(x0.dtype), 'jac_xf_p': su.to(x0.dtype)}
        x0 = np.asarray(x0, dtype=np.float64)
        lead = x0.shape[:-1]
        p = np.broadcast_to(np.asarray(p, dtype=np.float64), lead + (self.nu,)).reshape(-1, self.nu)
        x0 = x0.reshape(-1, self.nx)
        xf, m = self._flow(x0, p, rtol, atol)
        cols = []
        for j in range(self.nx + self.nu):
            v = x0 if j < self.nx else p
            k = j if j < self.nx else j - self.nx
            h = 6.0554544523933395e-6 * np.maximum(np.abs(v[..., k]), 1.0)
            h = (v[..., k] + h) - v[..., k]
            e = np.zeros(v.shape[-1]); e[k] = 1.0
            dv = h[..., None] * e
            plus = self._extrapolated_per_pair(x0 + dv, p, m) if j < self.nx else self._extrapolated_per_pair(x0, p + dv, m)
            minus = self._extrapolated_per_pair(x0 - dv, p, m) if j < self.nx else self._extrapolated_per_pair(x0, p - dv, m)
            cols.append((plus - minus) / (2.0 * h[..., None]))
        J = np.stack(cols, -1).reshape(lead + (self.nx, self.nx + self.nu))
        return {'xf': xf.reshape(lead + (self.nx,)), 'jac_xf_x0': J[..., :self.nx], 'jac_xf_p': J[..., self.nx:]}

    def g_func(self, x, u):
        return {'g': np.asarray(x, dtype=np.float64)}

    def dg_func(self, x, u):
        x = np.asarray(x)
        lead = x.shape[:-1]
        return {'dgdx': np.broadcast_to(np.eye(self.nx), lead + (self.nx, self.nx)).copy(),
                'dgdu': np.zeros(lead + (self.nx, self.nu))}

    def _build_jacobians(self):
        import sympy as sp
        X = sp.symbols('x0:%d' % self.nx)
        U = sp.symbols('u0:%d' % self.nu)
        F = sp.Matrix(_F[self.system](X, U, self.consts, sp))
        self._jac = (sp.lambdify([X, U], F.jacobian(X), 'numpy'), sp.lambdify([X, U], F.jacobian(U), 'numpy'))

    def df_func(self, x, u):
        """Jacobians of f: dfdx (..., nx, nx), dfdu (..., nx, nu).  Device tensors ([N, nx], [N, nu]) are
        linearised by the batched HIP kernel (``env.linearize``); host arrays by the sympy closed forms."""
        if _is_device_tensor(x):
            A, B = self._backend().linearize(x, u)
            return {'dfdx': A, 'dfdu': B}
        if self._jac is None:
            self._build_jacobians()
        x = np.asarray(x, dtype=np.float64)
        u = np.asarray(u, dtype=np.float64)
        lead = x.shape[:-1]
        xf, uf = x.reshape(-1, self.nx), u.reshape(-1, self.nu)
        A = np.stack([np.asarray(self._jac[0](xi, ui), dtype=np.float64) for xi, ui in zip(xf, uf)])
        B = np.stack([np.asarray(self._jac[1](xi, ui), dtype=np.float64) for xi, ui in zip(xf, uf)])
        return {'dfdx': A.reshape(lead + (self.nx, self.nx)), 'dfdu': B.reshape(lead + (self.nx, self.nu))}

    # -- linearised model: symbolic_systems.py:93-112 ------------------------------------------
    def fc_linear_func(self, x_eval, u_eval, x, u):
        """``f(x, u) + dfdx (x_eval - x) + dfdu (u_eval - u)`` (:95-101)."""
        if _is_device_tensor(x_eval) or _is_device_tensor(x):
            return {'f_linear': self._backend().model_linear_dynamics(x_eval, u_eval, x, u)}
        x_eval, u_eval = np.asarray(x_eval, dtype=np.float64), np.asarray(u_eval, dtype=np.float64)
        x, u = np.asarray(x, dtype=np.float64), np.asarray(u, dtype=np.float64)
        d = self.df_func(x, u)
        return {'f_linear': self._f(x, u) + np.einsum('...ij,...j->...i', d['dfdx'], x_eval - x)
                + np.einsum('...ij,...j->...i', d['dfdu'], u_eval - u)}

    def fd_linear_func(self, x0, p):
        """The linearised model integrated over ``dt`` from ``x0`` (= x_eval); ``p = [u_eval; x; u]`` as the CasADi
        integrator takes it (:102-107).  The ODE is affine, so its flow comes from one matrix exponential (the reference
        integrates it with CVODES)."""
        nx, nu = self.nx, self.nu
        if _is_device_tensor(x0):
            return {'xf': self._backend().model_linear_step(x0, p[..., :nu], p[..., nu:nu + nx], p[..., nu + nx:])}
        import scipy.linalg
        x0, p = np.asarray(x0, dtype=np.float64), np.asarray(p, dtype=np.float64)
        u_eval, x, u = p[..., :nu], p[..., nu:nu + nx], p[..., nu + nx:]
        d = self.df_func(x, u)
        c = self.fc_linear_func(np.zeros_like(x), u_eval, x, u)['f_linear']
        lead = x0.shape[:-1]
        A2, c2, z2 = d['dfdx'].reshape(-1, nx, nx), c.reshape(-1, nx), x0.reshape(-1, nx)
        out = np.empty_like(z2)
        for k in range(z2.shape[0]):
            M = np.zeros((nx + 1, nx + 1))
            M[:nx, :nx], M[:nx, nx] = A2[k], c2[k]
            E = scipy.linalg.expm(M * self.dt)
            out[k] = E[:nx, :nx] @ z2[k] + E[:nx, nx]
        return {'xf': out.reshape(lead + (nx,))}

    def g_linear_func(self, x_eval, u_eval, x, u):
        """``y + dgdx (x_eval - x) + dgdu (u_eval - u)`` with ``y = x`` (:108-112): the observation model is the identity."""
        if _is_device_tensor(x_eval):
            return {'g_linear': x + (x_eval - x)}
        x_eval, x = np.asarray(x_eval, dtype=np.float64), np.asarray(x, dtype=np.float64)
        return {'g_linear': x + (x_eval - x)}

    # -- cost: quadrotor.py:682, symbolic_systems.py:114-123 ------------------------------------
    def loss(self, x, u, Xr, Ur, Q, R):
        if _is_device_tensor(x):
            return self._backend().quadratic_loss(x, u, Xr, Ur, Q, R)
        x, u = np.asarray(x, dtype=np.float64), np.asarray(u, dtype=np.float64)
        ex = x - np.asarray(Xr, dtype=np.float64)
        eu = u - np.asarray(Ur, dtype=np.float64)
        Q, R = np.asarray(Q, dtype=np.float64), np.asarray(R, dtype=np.float64)
        Qs, Rs = 0.5 * (Q + Q.T), 0.5 * (R + R.T)
        l = 0.5 * np.einsum('...i,ij,...j->...', ex, Q, ex) + 0.5 * np.einsum('...i,ij,...j->...', eu, R, eu)
        lead = ex.shape[:-1]
        return {'l': l,
                'l_x': ex @ Qs.T, 'l_u': eu @ Rs.T,
                'l_xx': np.broadcast_to(Qs, lead + Qs.shape).copy(),
                'l_uu': np.broadcast_to(Rs, lead + Rs.shape).copy(),
                'l_xu': np.zeros(lead + (self.nx, self.nu))}


# the reference's class name (symbolic_systems.py:8): env.symbolic is an instance of it
SymbolicModel = AnalyticModel
