#!/usr/bin/env python
"""Generate tests/golden/models.npz: frozen outputs of the oracle for the next-row queries --
prior-model evaluation and linearisation (oracle/ref_model.py, oracle/ref_linearize.py), the quadratic loss, and the
running mean / std normalisers (oracle/ref_normalization.py).  As for make_golden.py these are outputs of OUR
restatement (the reference's modules need casadi / gym, absent here); they freeze it against drift
(tests/test_oracle_pinning.py::test_golden_models_match_oracle) and the GPU tests replay them through the HIP kernels
(tests/test_linearize.py::test_golden_models_on_device).  Run from the repo root:  python tests/golden/make_golden_models.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_linearize, ref_model, ref_normalization       # noqa: E402

SYSTEMS = {'q1': 1, 'q2': 2, 'q3': 3, 'cp': 'cartpole'}
F = {1: ref_model.f_quad1d, 2: ref_model.f_quad2d, 3: ref_model.f_quad3d, 'cartpole': ref_model.f_cartpole}
DT = 1.0 / 50


def build():
    rng = np.random.default_rng(424242)
    out = {}
    for tag, system in SYSTEMS.items():
        nx, nu = ref_model.NX[system], ref_model.NU[system]
        p = ref_model.CartPoleParams() if system == 'cartpole' else ref_model.QuadParams()
        x = (rng.normal(size=(16, nx)) * 0.5).astype(np.float32).astype(np.float64)
        if system == 'cartpole':
            u = rng.uniform(-10, 10, size=(16, nu)).astype(np.float32).astype(np.float64)
        else:
            lo, hi = ref_model.action_bounds(nu)
            u = rng.uniform(lo, hi, size=(16, nu)).astype(np.float32).astype(np.float64)
        A = np.zeros((16, nx, nx)); B = np.zeros((16, nx, nu)); f = np.zeros((16, nx)); xn = np.zeros((16, nx))
        for k in range(16):
            A[k], B[k] = ref_linearize.df(system, x[k], u[k], p)
            f[k] = F[system](list(x[k]), list(u[k]), p)
            xn[k] = ref_model.rk4_step(F[system], list(x[k]), list(u[k]), p, DT)
        Q = rng.normal(size=(nx, nx)).astype(np.float32).astype(np.float64)
        R = rng.normal(size=(nu, nu)).astype(np.float32).astype(np.float64)
        xr = rng.normal(size=nx).astype(np.float32).astype(np.float64)
        ur = rng.normal(size=nu).astype(np.float32).astype(np.float64)
        loss = [ref_linearize.quadratic_loss(x[k], u[k], xr, ur, Q, R) for k in range(16)]
        out.update({tag + '_x': x, tag + '_u': u, tag + '_A': A, tag + '_B': B, tag + '_f': f, tag + '_xnext': xn,
                    tag + '_Q': Q, tag + '_R': R, tag + '_xr': xr, tag + '_ur': ur,
                    tag + '_l': np.array([v['l'] for v in loss]), tag + '_lx': np.stack([v['l_x'] for v in loss]),
                    tag + '_lu': np.stack([v['l_u'] for v in loss])})
    # normaliser streams: three batches through one tracker
    norm = ref_normalization.MeanStdNormalizer(shape=(5,), clip=3.0)
    rnorm = ref_normalization.RewardStdNormalizer(gamma=0.9, clip=5.0)
    for b, n in enumerate((37, 64, 200)):
        xb = (rng.normal(size=(n, 5)) * [0.1, 1, 10, 3, 0.5] + [5, 0, -20, 1, 0]).astype(np.float32).astype(np.float64)
        out['norm_x%d' % b] = xb
        out['norm_y%d' % b] = norm(xb)
        out['norm_mean%d' % b], out['norm_var%d' % b] = norm.rms.mean.copy(), norm.rms.var.copy()
    for b in range(6):
        rew = rng.normal(-1, 2, size=50).astype(np.float32).astype(np.float64)
        done = rng.uniform(size=50) < 0.1
        out['rnorm_rew%d' % b], out['rnorm_done%d' % b] = rew, done
        out['rnorm_out%d' % b] = rnorm(rew, done)
        out['rnorm_ret%d' % b], out['rnorm_var%d' % b] = rnorm.ret.copy(), np.array(rnorm.rms.var)
    return out


if __name__ == '__main__':
    data = build()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'models.npz'), **data)
    print('wrote models.npz with %d arrays' % len(data))
