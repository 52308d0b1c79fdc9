#!/usr/bin/env python
"""CPU study for VERDICT r1 item 5 ("a cheaper accuracy-target mode"): which state planes must carry a float32 residual
(lo plane) for the open-loop 1000-step drift to stay <= 1e-5?

Scenario = tests/test_parity_gpu.py::test_precise_mode_open_loop_1000_step_drift (16 envs, PD feedback evaluated on the
float64 truth, the same float32 actions replayed on the candidate).  The candidate integrates the same RK4 in float64
but ROUNDS the chosen state components to float32 after every step (what storing only a hi plane does); optionally the
translational chain (positions, velocities) is also EVALUATED in float32.  Upper bounds on what the device could reach with
that storage scheme.  Bytes = 8 nx + 4 nu + 13 + 8 * (#lo planes).
Run: python tools/precise_mode_study.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_model as rm

P = rm.QuadParams()
KS2 = P.l / np.sqrt(2.0)
GAMMA = P.km / P.kf


def f(x, u, trans32=False):
    """quadrotor.py:649-674, vectorised over envs (rows)."""
    ph, th, ps, p, q, r = (x[:, j] for j in range(6, 12))
    T = u.sum(1) / P.m
    sph, cph, sth, cth, sps, cps = np.sin(ph), np.cos(ph), np.sin(th), np.cos(th), np.sin(ps), np.cos(ps)
    ax, ay, az = (cps * sth * cph + sps * sph) * T, (sps * sth * cph - cps * sph) * T, cth * cph * T - P.g
    if trans32:
        ax, ay, az = (v.astype(np.float32).astype(np.float64) for v in (ax, ay, az))
    mx = KS2 * (u[:, 0] + u[:, 1] - u[:, 2] - u[:, 3])
    my = KS2 * (-u[:, 0] + u[:, 1] + u[:, 2] - u[:, 3])
    mz = GAMMA * (u[:, 0] - u[:, 1] + u[:, 2] - u[:, 3])
    tth = sth / cth
    return np.stack([x[:, 1], ax, x[:, 3], ay, x[:, 5], az,
                     p + sph * tth * q + cph * tth * r, cph * q - sph * r, (sph * q + cph * r) / cth,
                     (mx - (q * P.izz * r - r * P.iyy * q)) / P.ixx, (my - (r * P.ixx * p - p * P.izz * r)) / P.iyy,
                     (mz - (p * P.iyy * q - q * P.ixx * p)) / P.izz], axis=1)


def rk4(x, u, dt, trans32=False):
    k1 = f(x, u, trans32); k2 = f(x + dt / 2 * k1, u, trans32); k3 = f(x + dt / 2 * k2, u, trans32); k4 = f(x + dt * k3, u, trans32)
    return x + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)


def motor(a):
    lo, hi = rm.action_bounds(4)
    t = np.clip(a, lo, hi)
    pwm = np.clip((np.sqrt(np.maximum(t, 0) / P.kf) - P.pwm2rpm_const) / P.pwm2rpm_scale, P.min_pwm, P.max_pwm)
    return P.kf * (P.pwm2rpm_scale * pwm + P.pwm2rpm_const) ** 2


def pd(xs, hover):
    k_arm, gamma = KS2, GAMMA
    th_des = np.clip(-0.25 * (xs[:, 0] - 0.2) - 0.35 * xs[:, 1], -0.3, 0.3)
    ph_des = np.clip(0.25 * (xs[:, 2] + 0.1) + 0.35 * xs[:, 3], -0.3, 0.3)
    thrust = hover * 4 * (1 + 1.5 * (1.0 - xs[:, 4]) - 0.8 * xs[:, 5])
    mx = 2e-3 * (ph_des - xs[:, 6]) - 4e-4 * xs[:, 9]
    my = 2e-3 * (th_des - xs[:, 7]) - 4e-4 * xs[:, 10]
    mz = -1e-3 * xs[:, 8] - 3e-4 * xs[:, 11]
    return np.stack([thrust / 4 + mx / (4 * k_arm) - my / (4 * k_arm) + mz / (4 * gamma), thrust / 4 + mx / (4 * k_arm) + my / (4 * k_arm) - mz / (4 * gamma),
                     thrust / 4 - mx / (4 * k_arm) + my / (4 * k_arm) + mz / (4 * gamma), thrust / 4 - mx / (4 * k_arm) - my / (4 * k_arm) - mz / (4 * gamma)], axis=1)


def run(lo_planes, trans32, steps=1000, seeds=(5, 6, 7)):
    worst_at = {100: 0.0, 300: 0.0, 1000: 0.0}
    for seed in seeds:
        rng = np.random.default_rng(seed)
        n = 16
        x0 = np.zeros((n, 12)); x0[:, [0, 2, 4]] = [0.2, -0.1, 1.0]
        x0 = (x0 + rng.normal(size=(n, 12)) * 0.05).astype(np.float32).astype(np.float64)
        truth, cand = x0.copy(), x0.copy()
        hover = P.m * P.g / 4
        hi_only = [j for j in range(12) if j not in lo_planes]
        worst = 0.0
        for k in range(steps):
            a = pd(truth, hover).astype(np.float32).astype(np.float64)
            u = motor(a)
            truth = rk4(truth, u, 1 / 60)
            cand = rk4(cand, u, 1 / 60, trans32)
            cand[:, hi_only] = cand[:, hi_only].astype(np.float32).astype(np.float64)      # no residual plane for these
            # hi + lo pair of float32 = 48 significant bits
            worst = max(worst, float(np.max(np.abs(cand - truth) / np.maximum(np.abs(truth), 1.0))))
            if k + 1 in worst_at:
                worst_at[k + 1] = max(worst_at[k + 1], worst)
    return worst_at


if __name__ == '__main__':
    ATT, VEL, POS = [6, 7, 8, 9, 10, 11], [1, 3, 5], [0, 2, 4]
    rows = [('all 12 lo planes (shipped precise mode)', list(range(12)), False),
            ('attitude + rates lo (6 planes); x, v hi only', ATT, False),
            ('attitude + rates + velocities lo (9 planes); positions hi only', ATT + VEL, False),
            ('attitude + rates + velocities lo (9), translational derivative rounded to f32', ATT + VEL, True),
            ('attitude + rates lo (6), translational derivative rounded to f32', ATT, True),
            ('no lo plane (fp64 arithmetic, fp32 storage)', [], False)]
    print('%-86s %6s  %9s %9s %9s' % ('storage scheme', 'B/step', '@100', '@300', '@1000'))
    for label, lo, t32 in rows:
        at = run(lo, t32)
        print('%-86s %6d  %9.1e %9.1e %9.1e' % (label, 125 + 8 * len(lo), at[100], at[300], at[1000]))
