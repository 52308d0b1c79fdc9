#!/usr/bin/env python
"""CPU study for VERDICT r1 item 5 ("a cheaper accuracy-target mode"): which state planes must carry a float32 residual
(lo plane) for the open-loop 1000-step drift to stay <= 1e-5?

Scenario = tests/test_parity_gpu.py::test_precise_mode_open_loop_1000_step_drift (16 envs, PD feedback evaluated on the
float64 truth, the same float32 actions replayed on the candidate).  The candidate integrates the same RK4 in float64
but ROUNDS the chosen state components to float32 after every step (what storing only a hi plane does); optionally the
translational chain (positions, velocities) is also EVALUATED in float32.  Upper bounds on what the device could reach with
that storage scheme.  Bytes = 8 nx + 4 nu + 13 + 8 * (#lo planes).
Round 3 adds the ARITHMETIC question (`python tests/studies/precise_mode_study.py --arithmetic`): with residual planes for attitude + rates,
which parts of the derivative may be evaluated in float32?  Two scenarios -- the PD hover above and the benchmarked workload (C4: circle
tracking under a PD tracking law, 64 envs).  The emulation reproduces the device: the first attitude-mode attempt of round 3 (float32
acceleration products) measured 3.2e-5 on the GPU at 1000 steps of circle tracking and 3.2e-5 here.
Run: python tests/studies/precise_mode_study.py [--arithmetic]        (CPU only; lives under tests/ because it imports the oracle)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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


# ---- round 3: which arithmetic of the derivative has to be double --------------------------------------------------------------
F32 = np.float32


def f_parts(x, u, trig32=False, kin32=False, acc32=False, torque32=False, bias=0.0, acc_df=False):
    """The derivative with chosen parts in float32 (inputs rounded, float32 operations); everything else float64.  The translational
    derivative is rounded to float32 at the end in every variant (the attitude mode stores positions / velocities in float32).
    bias: a constant relative error (1 + bias) on sin, (1 - bias) on cos."""
    c = lambda v: v.astype(F32)
    T = u.sum(1) / P.m
    if torque32:
        uf = c(u)
        mx, my, mz = (F32(KS2) * (uf[:, 0] + uf[:, 1] - uf[:, 2] - uf[:, 3])).astype(np.float64), \
                     (F32(KS2) * (-uf[:, 0] + uf[:, 1] + uf[:, 2] - uf[:, 3])).astype(np.float64), \
                     (F32(GAMMA) * (uf[:, 0] - uf[:, 1] + uf[:, 2] - uf[:, 3])).astype(np.float64)
    else:
        mx = KS2 * (u[:, 0] + u[:, 1] - u[:, 2] - u[:, 3]); my = KS2 * (-u[:, 0] + u[:, 1] + u[:, 2] - u[:, 3]); mz = GAMMA * (u[:, 0] - u[:, 1] + u[:, 2] - u[:, 3])
    ph, th, ps, p, q, r = (x[:, j] for j in range(6, 12))
    trig = (lambda a: (np.sin(c(a)).astype(np.float64), np.cos(c(a)).astype(np.float64))) if trig32 else (lambda a: (np.sin(a), np.cos(a)))
    (sph, cph), (sth, cth), (sps, cps) = trig(ph), trig(th), trig(ps)
    if bias:
        sph, sth, sps, cph, cth, cps = sph * (1 + bias), sth * (1 + bias), sps * (1 + bias), cph * (1 - bias), cth * (1 - bias), cps * (1 - bias)
    if acc_df:
        # float32 products; T/m and g enter the VERTICAL channel as double-float pairs (hi + lo), each FMA rounding a small result once
        fma32 = lambda a, b, cc: (a.astype(np.float64) * b.astype(np.float64) + cc).astype(F32)
        Thi = c(T); Tlo = c(T - Thi.astype(np.float64)); ghi = F32(P.g); glo = F32(P.g - np.float64(ghi))
        shc, chc = c(sth) * c(cph), c(cth) * c(cph)
        ax = fma32(c(cps), shc, (c(sps) * c(sph)).astype(np.float64)) * Thi
        ay = fma32(c(sps), shc, -(c(cps) * c(sph)).astype(np.float64)) * Thi
        az = fma32(chc, Thi, -np.float64(ghi)) + fma32(chc, Tlo, -np.float64(glo))
    elif acc32:
        ax, ay, az = (c(cps) * c(sth) * c(cph) + c(sps) * c(sph)) * c(T), (c(sps) * c(sth) * c(cph) - c(cps) * c(sph)) * c(T), c(cth) * c(cph) * c(T) - F32(P.g)
    else:
        ax, ay, az = (cps * sth * cph + sps * sph) * T, (sps * sth * cph - cps * sph) * T, cth * cph * T - P.g
    ax, ay, az = (v.astype(F32).astype(np.float64) for v in (ax, ay, az))
    if kin32:
        rc = F32(1) / c(cth); w = c(sph) * c(q) + c(cph) * c(r)
        rot = np.stack([(c(sth) * rc * w).astype(np.float64) + p, (c(cph) * c(q) - c(sph) * c(r)).astype(np.float64), (w * rc).astype(np.float64)], axis=1)
    else:
        tth = sth / cth
        rot = np.stack([p + sph * tth * q + cph * tth * r, cph * q - sph * r, (sph * q + cph * r) / cth], axis=1)
    rates = np.stack([(mx - (q * P.izz * r - r * P.iyy * q)) / P.ixx, (my - (r * P.ixx * p - p * P.izz * r)) / P.iyy, (mz - (p * P.iyy * q - q * P.ixx * p)) / P.izz], axis=1)
    return np.concatenate([np.stack([x[:, 1], ax, x[:, 3], ay, x[:, 5], az], axis=1), rot, rates], axis=1)


def pd_tracking(xs, goal, hover):
    th_des = np.clip(-0.25 * (xs[:, 0] - goal[0]) - 0.35 * (xs[:, 1] - goal[1]), -0.3, 0.3)
    ph_des = np.clip(0.25 * (xs[:, 2] - goal[2]) + 0.35 * (xs[:, 3] - goal[3]), -0.3, 0.3)
    thrust = hover * 4 * (1 + 1.5 * (goal[4] - xs[:, 4]) - 0.8 * (xs[:, 5] - goal[5]))
    mx = 2e-3 * (ph_des - xs[:, 6]) - 4e-4 * xs[:, 9]; my = 2e-3 * (th_des - xs[:, 7]) - 4e-4 * xs[:, 10]; mz = -1e-3 * xs[:, 8] - 3e-4 * xs[:, 11]
    return np.stack([thrust / 4 + mx / (4 * KS2) - my / (4 * KS2) + mz / (4 * GAMMA), thrust / 4 + mx / (4 * KS2) + my / (4 * KS2) - mz / (4 * GAMMA),
                     thrust / 4 - mx / (4 * KS2) + my / (4 * KS2) + mz / (4 * GAMMA), thrust / 4 - mx / (4 * KS2) - my / (4 * KS2) - mz / (4 * GAMMA)], axis=1)


def run_arithmetic(scenario, steps=1000, **parts):
    """Worst relative state error at 100 / 300 / 1000 steps; residual planes for states 6..11, the same float32 actions on both sides."""
    lo_planes = [6, 7, 8, 9, 10, 11]
    hi_only = [j for j in range(12) if j not in lo_planes]
    hover = P.m * P.g / 4
    if scenario == 'hover':
        rng = np.random.default_rng(5); n = 16
        x0 = np.zeros((n, 12)); x0[:, [0, 2, 4]] = [0.2, -0.1, 1.0]
        x0 = (x0 + rng.normal(size=(n, 12)) * 0.05).astype(F32).astype(np.float64)
        act = lambda xs, k: pd(xs, hover)
    else:
        from oracle.ref_env import RefEnv
        xg = np.asarray(RefEnv(system=3, task='traj_tracking', cost='quadratic', ctrl_freq=60, pyb_freq=240, done_on_out_of_bound=False,
                               episode_len_sec=20).X_GOAL)
        n = 64
        x0 = (xg[0][None, :] + np.random.default_rng(6).normal(size=(n, 12)) * 0.03).astype(F32).astype(np.float64)
        act = lambda xs, k: pd_tracking(xs, xg[min(k, len(xg) - 1)], hover)
    step = lambda x, u, fn: x + (1 / 60) / 6 * (lambda k1: (lambda k2: (lambda k3: k1 + 2 * k2 + 2 * k3 + fn(x + (1 / 60) * k3, u))(fn(x + (1 / 120) * k2, u)))(fn(x + (1 / 120) * k1, u)))(fn(x, u))
    truth, cand, worst, at = x0.copy(), x0.copy(), 0.0, {}
    for k in range(steps):
        u = motor(act(truth, k).astype(F32).astype(np.float64))
        truth = step(truth, u, f)
        cand = step(cand, u, lambda x, uu: f_parts(x, uu, **parts))
        cand[:, hi_only] = cand[:, hi_only].astype(F32).astype(np.float64)
        worst = max(worst, float(np.max(np.abs(cand - truth) / np.maximum(np.abs(truth), 1.0))))
        if k + 1 in (100, 300, 1000):
            at[k + 1] = worst
    return at


ARITHMETIC_ROWS = [
    ('everything in double (round 2\'s attitude mode)', {}),
    ('sin / cos in float32 on the float32-rounded angle', dict(trig32=True)),
    ('Euler-rate kinematics in float32', dict(kin32=True)),
    ('acceleration products R(:,2) T/m - g in float32', dict(acc32=True)),
    ('torques from float32 thrusts', dict(torque32=True)),
    ('constant bias 6e-8 on sin / cos (double arithmetic)', dict(bias=6e-8)),
    ('float32 sin / cos + kinematics, double torques / rates / acceleration products', dict(trig32=True, kin32=True)),
    ('SHIPPED: ... and the acceleration in float32 with T/m and g as double-float pairs in the vertical channel', dict(trig32=True, kin32=True, acc_df=True)),
    ('first attempt: float32 sin / cos + kinematics + acceleration products', dict(trig32=True, kin32=True, acc32=True)),
]


if __name__ == '__main__':
    if '--arithmetic' in sys.argv:
        print('%-92s %-30s %-30s' % ('arithmetic of the derivative (residual planes for states 6..11)', 'hover @100 / @300 / @1000', 'C4 circle tracking @100 / @300 / @1000'))
        for label, parts in ARITHMETIC_ROWS:
            h, c4 = run_arithmetic('hover', **parts), run_arithmetic('c4', **parts)
            print('%-92s %-30s %-30s' % (label, ' / '.join('%.1e' % h[k] for k in (100, 300, 1000)), ' / '.join('%.1e' % c4[k] for k in (100, 300, 1000))))
        sys.exit(0)
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
