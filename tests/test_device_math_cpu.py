"""CPU emulation of the device's float32 sin / cos (safe_control_gym_amd/csrc/scg_math.h::sincos_f): the same sequence of float32
operations -- magic-number quadrant, two-piece Cody-Waite reduction, the two minimax polynomials, quadrant fix-up on the low bits of
the magic sum -- evaluated with NumPy, its constants READ from the header, against float64 sin / cos.  Pins the accuracy DESIGN.md
section 5 states (max abs error 9.3e-8 for |x| <= 1000) to the constants and the reduction actually shipped."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


def _constants():
    text = open(os.path.join(ROOT, 'safe_control_gym_amd', 'csrc', 'scg_math.h')).read()
    vals = dict(re.findall(r'\b(k[A-Z][A-Za-z0-9]*) = (-?[0-9.]+(?:e-?[0-9]+)?)f', text))
    need = ('kTwoOverPi', 'kPio2Hi', 'kPio2Lo', 'kRintMagic', 'kS3', 'kS2', 'kS1', 'kS0', 'kC2', 'kC1', 'kC0')
    assert all(k in vals for k in need), sorted(vals)
    return {k: F(float(vals[k])) for k in need}


def _fma(a, b, c):
    """float32 fma: the product of two float32 is exact in float64; the sum is rounded to float64 and then to float32 (the
    double rounding can differ from a true fma by half an ulp in rare ties: immaterial for an error BOUND)."""
    return (np.asarray(a, dtype=np.float64) * np.asarray(b, dtype=np.float64) + np.asarray(c, dtype=np.float64)).astype(F)


def sincos_f(x, k):
    x = x.astype(F)
    t = _fma(x, k['kTwoOverPi'], k['kRintMagic'])
    kf = (t - k['kRintMagic']).astype(F)
    r = _fma(kf, k['kPio2Hi'], x)
    r = _fma(kf, k['kPio2Lo'], r)
    r2 = (r * r).astype(F)
    ps = _fma(_fma(_fma(r2, k['kS3'], k['kS2']), r2, k['kS1']), r2, k['kS0'])
    sr = _fma((r * r2).astype(F), ps, r)
    pc = _fma(_fma(r2, k['kC2'], k['kC1']), r2, k['kC0'])
    cr = _fma((r2 * r2).astype(F), pc, _fma(r2, F(-0.5), F(1.0)))
    kb = t.view(np.uint32)                                   # k mod 4 sits in the two low mantissa bits of the magic sum
    s = np.where(kb & 1, cr, sr)
    c = np.where(kb & 1, sr, cr)
    return np.where(kb & 2, -s, s), np.where((kb + 1) & 2, -c, c)


def test_float32_sincos_accuracy_and_quadrants():
    k = _constants()
    assert k['kRintMagic'] == F(1.5 * 2 ** 23)
    rng = np.random.default_rng(0)
    for bound in (3.2, 100.0, 1000.0):
        x = rng.uniform(-bound, bound, size=1_000_000).astype(F)
        s, c = sincos_f(x, k)
        assert np.abs(s.astype(np.float64) - np.sin(x.astype(np.float64))).max() <= 9.5e-8, bound
        assert np.abs(c.astype(np.float64) - np.cos(x.astype(np.float64))).max() <= 9.5e-8, bound
    # exact values and every quadrant, both signs
    x = (np.arange(-16, 17) * (np.pi / 4)).astype(F)
    s, c = sincos_f(x, k)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() <= 1.2e-7 and np.abs(c - np.cos(x.astype(np.float64))).max() <= 1.2e-7
    z = sincos_f(np.zeros(1, dtype=F), k)
    assert z[0][0] == 0.0 and z[1][0] == 1.0
