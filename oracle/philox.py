"""Oracle: Philox4x32-10 counter-based RNG in NumPy integer arithmetic (bit-exact).

Test infrastructure -- see ``oracle/__init__.py``.

The reference draws reset randomisation from ``gym.utils.seeding.np_random`` (MT19937 behind a
SHA-512 seed hash; envs/benchmark_env.py:231,297-307), one RandomState per env with
``seed + rank`` (envs/env_wrappers/vectorized_env/__init__.py:28-33).  A per-env Mersenne Twister
is 2.5 KB of state per env and is not reproducible without gym; the device engine instead uses
the published Philox4x32-10 generator (Salmon et al., SC'11 -- Random123), keyed by the seed and
countered by (global env index, reset serial).  This file restates that algorithm so that parity
tests can give the float64 oracle the *same* reset draws as the HIP kernel.  Pinned by the
Random123 known-answer vectors (tests/test_oracle_philox.py).

Draw layout (shared with safe_control_gym_amd/csrc/scg_kernels.hip):
    key = (seed & 0xffffffff, seed >> 32)
    ctr = (env_index & 0xffffffff, env_index >> 32, reset_serial, j)     j = 0,1,2
    words 4j .. 4j+3 -> state dims 4j .. 4j+3
    u = (word >> 8) * 2**-24  in [0, 1);  offset = low + (high - low) * u
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32, key: (..., 2) uint32 -> (..., 4) uint32."""
    c = [np.asarray(ctr[..., i], dtype=np.uint32) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32)
    k1 = np.asarray(key[..., 1], dtype=np.uint32)
    with np.errstate(over='ignore'):
        for r in range(10):
            p0 = _M0 * c[0].astype(np.uint64)
            p1 = _M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            if r < 9:
                k0 = (k0 + _W0).astype(np.uint32)
                k1 = (k1 + _W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def uniform01(words):
    """(word >> 8) * 2^-24, exactly representable in float32 and float64."""
    return (np.asarray(words, dtype=np.uint32) >> np.uint32(8)).astype(np.float64) * (2.0 ** -24)


# scg_distribution (include/scg_rollout.h): the np.random.RandomState methods the reference may name in its yaml
# (envs/benchmark_env.py:297-307), as transforms of the Philox uniforms -- shared with csrc/scg_kernels.hip::draw_distribution
(UNIFORM, NORMAL, EXPONENTIAL, LAPLACE, LOGISTIC, TRIANGULAR, RANDINT, CHOICE, RAYLEIGH, GUMBEL, WEIBULL, PARETO, POWER, LOGNORMAL, CAUCHY,
 GAMMA, BETA, CHOICE_P) = range(18)
REJECT_WORD_INIT, REJECT_WORD_INERTIAL, GAMMA_ATTEMPTS = 256, 512, 8


def gamma_variate(a, ctx):
    """gamma(shape a, 1) by Marsaglia-Tsang on Philox blocks (seed, env, serial, word0 + attempt) -- the twin of
    csrc/scg_kernels.hip::gamma_variate (same words, same order, float64 arithmetic: an acceptance that falls within float32
    rounding of its threshold may differ from the device's)."""
    seed, env_index, serial, word0 = ctx
    ab = a + 1.0 if a < 1.0 else a
    d = ab - 1.0 / 3.0
    c = 1.0 / np.sqrt(9.0 * d)
    cand, boost = d, 0.5
    for k in range(GAMMA_ATTEMPTS):
        w = _uniform_block(seed, env_index, serial, word0 + k)
        u0, u1, u2 = w[0] + 2.0 ** -25, w[1], w[2] + 2.0 ** -25
        if k == 0:
            boost = w[3] + 2.0 ** -25
        x = np.sqrt(-2.0 * np.log(u0)) * np.cos(2.0 * np.pi * u1)
        v = (1.0 + c * x) ** 3
        if v > 0.0:
            cand = d * v
            if np.log(u2) < 0.5 * x * x + d - cand + d * np.log(v):
                break
    if a < 1.0:
        cand *= boost ** (1.0 / a)
    return cand


def draw(kind, u, u2, p, low=0.0, high=1.0, ctx=None):
    """One draw, float64.  u, u2 in [0, 1); p = (p0, p1, p2) as documented for scg_distribution; ctx = (seed, env index, serial,
    word0) for the rejection samplers."""
    uo = u + 2.0 ** -25
    if kind == GAMMA:
        return p[1] * gamma_variate(p[0], ctx)
    if kind == BETA:
        x = gamma_variate(p[0], ctx)
        y = gamma_variate(p[1], (ctx[0], ctx[1], ctx[2], ctx[3] + GAMMA_ATTEMPTS))
        return x / (x + y)
    if kind == CHOICE_P:                                     # low = P(p0), high = P(p0) + P(p1)
        return p[0] if u < low else (p[1] if u < high else p[2])
    if kind == NORMAL:
        return p[0] + p[1] * np.sqrt(-2.0 * np.log(uo)) * np.cos(2.0 * np.pi * u2)
    if kind == EXPONENTIAL:
        return -p[0] * np.log(1.0 - u)
    if kind == LAPLACE:
        d = u - 0.5
        return p[0] + np.copysign(p[1] * np.log(max(1.0 - 2.0 * abs(d), 2.0 ** -24)), d)   # the monotone inverse CDF (numpy's)
    if kind == LOGISTIC:
        return p[0] + p[1] * np.log(uo / (1.0 - uo))
    if kind == TRIANGULAR:
        base, ratio = p[2] - p[0], (p[1] - p[0]) / (p[2] - p[0])
        return p[0] + np.sqrt(u * base * (p[1] - p[0])) if u <= ratio else p[2] - np.sqrt((1.0 - u) * base * (p[2] - p[1]))
    if kind == RANDINT:
        return np.floor(p[0] + (p[1] - p[0]) * u)
    if kind == CHOICE:                                       # `low` carries the number of equiprobable values (1..3)
        return p[min(int(u * low), int(low) - 1)]
    if kind == RAYLEIGH:
        return p[0] * np.sqrt(-2.0 * np.log(1.0 - u))
    if kind == GUMBEL:
        return p[0] - p[1] * np.log(-np.log(uo))
    if kind == WEIBULL:
        return (-np.log(1.0 - u)) ** (1.0 / p[0])
    if kind == PARETO:
        return np.exp(-np.log(1.0 - u) / p[0]) - 1.0
    if kind == POWER:
        return uo ** (1.0 / p[0])
    if kind == LOGNORMAL:
        return np.exp(p[0] + p[1] * np.sqrt(-2.0 * np.log(uo)) * np.cos(2.0 * np.pi * u2))
    if kind == CAUCHY:
        return np.tan(np.pi * (uo - 0.5))
    return low + (high - low) * u


def _uniform_block(seed, env_index, reset_serial, word3):
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    ctr = np.array([env_index & 0xFFFFFFFF, (env_index >> 32) & 0xFFFFFFFF, reset_serial & 0xFFFFFFFF, word3], dtype=np.uint32)
    return uniform01(philox4x32_10(ctr, key))


def reset_offsets(seed, env_index, reset_serial, low, high, kinds=None, params=None):
    """Reset offsets for one env: (nx,) float64; low / high are (nx,) arrays (uniform dims).  ``kinds`` / ``params``
    ((nx,) ints, (nx, 3) floats) select other distributions per dim; their second uniforms come from blocks 4 + j."""
    low = np.asarray(low, dtype=np.float64)
    high = np.asarray(high, dtype=np.float64)
    nx = low.shape[0]
    general = kinds is not None and any(int(k) != UNIFORM for k in kinds)
    out = np.zeros(nx)
    for j in range((nx + 3) // 4):
        u = _uniform_block(seed, env_index, reset_serial, j)
        u2 = _uniform_block(seed, env_index, reset_serial, 4 + j) if general else np.zeros(4)
        for w in range(4):
            d = 4 * j + w
            if d < nx:
                out[d] = draw(int(kinds[d]), u[w], u2[w], params[d], low[d], high[d],
                              (seed, env_index, reset_serial, REJECT_WORD_INIT + 16 * d)) if general else low[d] + (high[d] - low[d]) * u[w]
    return out


def inertial_offsets(seed, env_index, reset_serial, low, high, kinds=None, params=None):
    """Offsets for the 4 inertial properties: Philox block 15 of the same (env, serial) counter (block 14: the second
    uniforms of non-uniform kinds)."""
    u = _uniform_block(seed, env_index, reset_serial, 15)
    low, high = np.asarray(low, dtype=np.float64), np.asarray(high, dtype=np.float64)
    if kinds is None or all(int(k) == UNIFORM for k in kinds):
        return low + (high - low) * u
    u2 = _uniform_block(seed, env_index, reset_serial, 14)
    return np.array([draw(int(kinds[q]), u[q], u2[q], params[q], low[q], high[q], (seed, env_index, reset_serial, REJECT_WORD_INERTIAL + 16 * q))
                     for q in range(4)])
