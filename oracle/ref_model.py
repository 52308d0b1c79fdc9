"""Oracle: prior dynamics model, fixed-step integrators and the motor map (float64).

Test infrastructure -- see ``oracle/__init__.py``.  Scalar, one environment at a
time, written with ``math`` so that every operation is an IEEE double.

Reference anchors (relative to /root/reference/safe_control_gym):
  * physical constants ......... envs/gym_pybullet_drones/assets/cf2x.urdf:5,12,13
                                 envs/gym_pybullet_drones/base_aviary.py:90 (g = 9.8)
  * continuous dynamics f(x,u) . envs/gym_pybullet_drones/quadrotor.py:594-676
  * rotation R = Rz Ry Rx ...... math_and_models/transformations.py:125-182
  * RK4 (ZOH input) ............ controllers/mpc/mpc_utils.py:79-101
  * motor map .................. envs/gym_pybullet_drones/quadrotor_utils.py:21-67
                                 envs/gym_pybullet_drones/base_aviary.py:412-435 (KF*rpm^2)
  * action bounds .............. envs/gym_pybullet_drones/quadrotor.py:726-733
"""
import math

import numpy as np

# --- assets/cf2x.urdf:5,12,13 -------------------------------------------------------------
CF2X = dict(
    M=0.03454, L=0.0397, IXX=1.4e-5, IYY=1.4e-5, IZZ=2.17e-5,
    KF=3.16e-10, KM=7.94e-12,
    PWM2RPM_SCALE=0.2685, PWM2RPM_CONST=4070.3, MIN_PWM=20000.0, MAX_PWM=65535.0,
    THRUST2WEIGHT_RATIO=2.25, GND_EFF_COEFF=11.36859, PROP_RADIUS=2.31348e-2, DRAG_COEFF_XY=9.1785e-7,
    DRAG_COEFF_Z=10.311e-7,                                          # assets/cf2x.urdf:5
)
GRAVITY_ACC = 9.8  # base_aviary.py:90

NX = {1: 2, 2: 6, 3: 12, 'cartpole': 4}
NU = {1: 1, 2: 2, 3: 4, 'cartpole': 1}


class QuadParams:
    """Physical parameters of one quadrotor (nominal = cf2x.urdf)."""

    def __init__(self, m=None, ixx=None, iyy=None, izz=None):
        self.m = CF2X['M'] if m is None else float(m)
        self.ixx = CF2X['IXX'] if ixx is None else float(ixx)
        self.iyy = CF2X['IYY'] if iyy is None else float(iyy)
        self.izz = CF2X['IZZ'] if izz is None else float(izz)
        self.g = GRAVITY_ACC
        self.l = CF2X['L']
        self.kf = CF2X['KF']
        self.km = CF2X['KM']
        self.pwm2rpm_scale = CF2X['PWM2RPM_SCALE']
        self.pwm2rpm_const = CF2X['PWM2RPM_CONST']
        self.min_pwm = CF2X['MIN_PWM']
        self.max_pwm = CF2X['MAX_PWM']


# --- quadrotor.py:594-676 -----------------------------------------------------------------
def f_quad1d(x, u, p, fext=(0.0, 0.0, 0.0)):
    """quadrotor.py:602  X_dot = [z_dot, T/m - g]."""
    z_dot = x[1]
    T = u[0]
    return [z_dot, T / p.m - p.g + fext[2] / p.m]


def f_quad2d(x, u, p, fext=(0.0, 0.0, 0.0)):
    """quadrotor.py:618-621."""
    x_dot, z_dot, theta, theta_dot = x[1], x[3], x[4], x[5]
    T1, T2 = u[0], u[1]
    return [
        x_dot,
        math.sin(theta) * (T1 + T2) / p.m + fext[0] / p.m,
        z_dot,
        math.cos(theta) * (T1 + T2) / p.m - p.g + fext[2] / p.m,
        theta_dot,
        p.l * (T2 - T1) / p.iyy / math.sqrt(2),
    ]


def rot_xyz(phi, theta, psi):
    """transformations.py:125-182: R = Rz(psi) @ Ry(theta) @ Rx(phi) as nested lists."""
    cz, sz = math.cos(psi), math.sin(psi)
    cy, sy = math.cos(theta), math.sin(theta)
    cx, sx = math.cos(phi), math.sin(phi)
    Rz = [[cz, -sz, 0.0], [sz, cz, 0.0], [0.0, 0.0, 1.0]]
    Ry = [[cy, 0.0, sy], [0.0, 1.0, 0.0], [-sy, 0.0, cy]]
    Rx = [[1.0, 0.0, 0.0], [0.0, cx, -sx], [0.0, sx, cx]]

    def mm(A, B):
        return [[sum(A[i][k] * B[k][j] for k in range(3)) for j in range(3)] for i in range(3)]
    return mm(mm(Rz, Ry), Rx)


def f_quad3d(x, u, p, fext=(0.0, 0.0, 0.0)):
    """quadrotor.py:624-674.  State (x,xd,y,yd,z,zd,phi,theta,psi,p,q,r), input (f1..f4)."""
    x_dot, y_dot, z_dot = x[1], x[3], x[5]
    phi, theta, psi = x[6], x[7], x[8]
    pp, q, r = x[9], x[10], x[11]
    f1, f2, f3, f4 = u
    R = rot_xyz(phi, theta, psi)
    thrust = f1 + f2 + f3 + f4
    # :664  oVdot = Rob @ [0,0,sum f]/m - [0,0,g]   (+ world-frame disturbance force, base_aviary.py:306-315)
    acc = [R[i][2] * thrust / p.m + fext[i] / p.m for i in range(3)]
    acc[2] -= p.g
    gamma = p.km / p.kf  # :634
    s2 = math.sqrt(2.0)
    Mb = [p.l / s2 * (f1 + f2 - f3 - f4),     # :667
          p.l / s2 * (-f1 + f2 + f3 - f4),    # :668
          gamma * (f1 - f2 + f3 - f4)]        # :669
    # :670  Jinv @ (Mb - skew(w) @ J @ w),  skew(w) v = w x v
    Jw = [p.ixx * pp, p.iyy * q, p.izz * r]
    cross = [q * Jw[2] - r * Jw[1], r * Jw[0] - pp * Jw[2], pp * Jw[1] - q * Jw[0]]
    rate_dot = [(Mb[0] - cross[0]) / p.ixx, (Mb[1] - cross[1]) / p.iyy, (Mb[2] - cross[2]) / p.izz]
    # :671-673
    sphi, cphi = math.sin(phi), math.cos(phi)
    tth, cth = math.tan(theta), math.cos(theta)
    ang_dot = [pp + sphi * tth * q + cphi * tth * r,
               cphi * q - sphi * r,
               sphi / cth * q + cphi / cth * r]
    # :674 stacking order
    return [x_dot, acc[0], y_dot, acc[1], z_dot, acc[2],
            ang_dot[0], ang_dot[1], ang_dot[2],
            rate_dot[0], rate_dot[1], rate_dot[2]]


class CartPoleParams:
    """CartPole parameters.  NOT in the reference fork: names follow
    tests/scripts/verbose_api.yaml:32-39; l is the *effective* (half) pole length."""

    def __init__(self, pole_length=0.5, pole_mass=0.1, cart_mass=1.0):
        self.l = float(pole_length)
        self.m = float(pole_mass)
        self.M = float(cart_mass)
        self.g = GRAVITY_ACC


def f_cartpole(x, u, p, fext=(0.0, 0.0, 0.0)):
    """Textbook (Barto/Florian) cart-pole, state (x, x_dot, theta, theta_dot), input force F.
    Parity unpinned: there is no CartPole env in /root/reference.
    An external (disturbance) force fext[0] acts on the cart along x, next to the input force."""
    x_dot, theta, theta_dot = x[1], x[2], x[3]
    Mm = p.m + p.M
    ml = p.m * p.l
    s, c = math.sin(theta), math.cos(theta)
    tmp = (u[0] + fext[0] + ml * theta_dot * theta_dot * s) / Mm
    theta_dd = (p.g * s - c * tmp) / (p.l * (4.0 / 3.0 - p.m * c * c / Mm))
    x_dd = tmp - ml * theta_dd * c / Mm
    return [x_dot, x_dd, theta_dot, theta_dd]


F_BY_SYSTEM = {1: f_quad1d, 2: f_quad2d, 3: f_quad3d, 'cartpole': f_cartpole}


# --- controllers/mpc/mpc_utils.py:79-101 --------------------------------------------------
def rk4_step(f, x, u, p, dt, fext=(0.0, 0.0, 0.0)):
    """k1..k4 with zero-order-hold input; x+ = x + dt/6 (k1 + 2 k2 + 2 k3 + k4)."""
    n = len(x)
    k1 = f(x, u, p, fext)
    k2 = f([x[i] + dt / 2 * k1[i] for i in range(n)], u, p, fext)
    k3 = f([x[i] + dt / 2 * k2[i] for i in range(n)], u, p, fext)
    k4 = f([x[i] + dt * k3[i] for i in range(n)], u, p, fext)
    return [x[i] + dt / 6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) for i in range(n)]


def euler_step(f, x, u, p, dt, fext=(0.0, 0.0, 0.0)):
    """Explicit Euler x+ = x + dt f(x,u).  (No in-tree spec: base_aviary.py:560-563 is dead
    code and semi-implicit; this is the plain textbook form, documented in DESIGN.md.)"""
    k1 = f(x, u, p, fext)
    return [x[i] + dt * k1[i] for i in range(len(x))]


INTEGRATORS = {'rk4': rk4_step, 'euler': euler_step}


# --- quadrotor.py:726-733 -----------------------------------------------------------------
def action_bounds(nu, p=None):
    """a_low/a_high = KF * (4/nu) * (S*PWM_{min,max} + C)^2, stored as float32 by
    gym.spaces.Box(dtype=float32) (quadrotor.py:731-733) -- so returned as float32-rounded
    float64 values, which is what np.clip sees at quadrotor.py:830."""
    p = p or QuadParams()
    n_motors = 4 / nu
    a_low = p.kf * n_motors * (p.pwm2rpm_scale * p.min_pwm + p.pwm2rpm_const) ** 2
    a_high = p.kf * n_motors * (p.pwm2rpm_scale * p.max_pwm + p.pwm2rpm_const) ** 2
    return float(np.float32(a_low)), float(np.float32(a_high))


# --- quadrotor_utils.py:21-67 + base_aviary.py:420 ------------------------------------------
def motor_map(thrust, p):
    """thrust (nu,) -> effective per-input thrust after cmd2pwm -> pwm2rpm -> KF*rpm^2.

    cmd2pwm: n_motor = 4 // nu; thrust = max(thrust, 0); pwm = (sqrt(thrust/n_motor/KF) - C)/S;
    clip to [PWM_min, PWM_max]; pwm2rpm: rpm = S*pwm + C; force per motor = KF*rpm^2
    (base_aviary.py:420).  The nu inputs drive n_motor identical motors each, so the
    effective thrust of input j is n_motor*KF*rpm_j^2 -- the quantity the analytic model's
    u_j stands for (1-D: total thrust; 2-D: T1 = motors 1&4, T2 = motors 2&3,
    quadrotor_utils.py:24-25,43-44).
    """
    nu = len(thrust)
    n_motor = 4 // nu
    out = []
    for T in thrust:
        T = max(T, 0.0)
        pwm = (math.sqrt(T / n_motor / p.kf) - p.pwm2rpm_const) / p.pwm2rpm_scale
        pwm = min(max(pwm, p.min_pwm), p.max_pwm)
        rpm = p.pwm2rpm_scale * pwm + p.pwm2rpm_const
        out.append(n_motor * p.kf * rpm * rpm)
    return out


# --- base_aviary.py:437-500: aero terms of Physics.PYB_GND / PYB_DRAG, restated for the analytic model -------------
# "Parity unpinned": the reference applies these forces inside PyBullet; here they enter the prior model as
# zero-order-hold inputs (drag from the state at the start of the control step, ground effect per integration sub-step).
def gnd_eff_h_clip():
    """base_aviary.py:141-147 (computed from the URDF mass)."""
    max_rpm2 = CF2X['THRUST2WEIGHT_RATIO'] * GRAVITY_ACC * CF2X['M'] / (4 * CF2X['KF'])
    max_thrust = 4 * CF2X['KF'] * max_rpm2
    return 0.25 * CF2X['PROP_RADIUS'] * math.sqrt(15 * max_rpm2 * CF2X['KF'] * CF2X['GND_EFF_COEFF'] / max_thrust)


def _rot_for(system, x):
    if system == 3:
        return rot_xyz(x[6], x[7], x[8])
    th = x[4] if system == 2 else 0.0
    c, s = math.cos(th), math.sin(th)
    return [[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]]


def _vel_of(system, x):
    if system == 1:
        return [0.0, 0.0, x[1]]
    if system == 2:
        return [x[1], 0.0, x[3]]
    return [x[1], x[3], x[5]]


def drag_force(system, x, u, p):
    """:478-496.  rpm_i from the per-motor thrust u_j / n_motor = KF rpm^2; world force = R (R (k * v))."""
    nu = len(u)
    n_motor = 4 // nu
    srpm = n_motor * sum(math.sqrt(max(uj, 0.0) / (n_motor * p.kf)) for uj in u)
    k = [-CF2X['DRAG_COEFF_XY'] * 2 * math.pi / 60 * srpm] * 2 + [-CF2X['DRAG_COEFF_Z'] * 2 * math.pi / 60 * srpm]
    v = _vel_of(system, x)
    R = _rot_for(system, x)
    w = [k[i] * v[i] for i in range(3)]
    b = [sum(R[i][j] * w[j] for j in range(3)) for i in range(3)]
    return [sum(R[i][j] * b[j] for j in range(3)) for i in range(3)]


def ground_effect_thrusts(system, x, u, p):
    """:437-476: thrust of propeller i scaled by 1 + GND_EFF_COEFF (PROP_RADIUS / (4 h_i))^2 while |roll|, |pitch| < pi/2."""
    a = p.l / math.sqrt(2.0)
    clip = gnd_eff_h_clip()
    gain = lambda h: 1.0 + CF2X['GND_EFF_COEFF'] * (CF2X['PROP_RADIUS'] / (4 * max(h, clip))) ** 2
    if system == 1:
        return [u[0] * gain(x[0])]
    R = _rot_for(system, x)
    if system == 2:
        if abs(x[4]) >= math.pi / 2:
            return list(u)
        return [u[0] * gain(x[2] + R[2][0] * a), u[1] * gain(x[2] - R[2][0] * a)]        # T1 at x = +a, T2 at x = -a
    if abs(x[6]) >= math.pi / 2 or abs(x[7]) >= math.pi / 2:
        return list(u)
    offs = [(a, a), (-a, a), (-a, -a), (a, -a)]                                         # cf2x.urdf:43-79
    return [u[i] * gain(x[4] + R[2][0] * ox + R[2][1] * oy) for i, (ox, oy) in enumerate(offs)]
