"""Distributions of the reset-time randomisation.

The reference draws every randomised quantity with ``getattr(self.np_random, info['distrib'])(*args, **kwargs)``
(/root/reference/safe_control_gym/envs/benchmark_env.py:297-307): any ``numpy.random.RandomState`` method name is legal in the
yaml.  On the device the draw is a transform of one or two Philox uniforms per quantity (``scg_distribution`` in
include/scg_rollout.h); this module maps the RandomState signatures of the supported methods onto it.  ``gamma`` (and ``standard_gamma``, ``chisquare``) and
``beta`` are bounded-iteration rejection samplers on Philox blocks of their own.  What stays out raises ``NotImplementedError`` naming
the method: array-valued draws (``permutation``, ``multivariate_normal`` ...), ``vonmises`` / ``f`` / ``standard_t`` ..., and ``choice``
over more than three values (the device record of a quantity holds three parameters).
"""
from safe_control_gym_amd import _capi

# method -> (kind, positional / keyword parameter names with RandomState's defaults)
_SIGNATURES = {
    'uniform': (_capi.SCG_RAND_UNIFORM, (('low', 0.0), ('high', 1.0))),
    'normal': (_capi.SCG_RAND_NORMAL, (('loc', 0.0), ('scale', 1.0))),
    'standard_normal': (_capi.SCG_RAND_NORMAL, ()),
    'exponential': (_capi.SCG_RAND_EXPONENTIAL, (('scale', 1.0),)),
    'laplace': (_capi.SCG_RAND_LAPLACE, (('loc', 0.0), ('scale', 1.0))),
    'logistic': (_capi.SCG_RAND_LOGISTIC, (('loc', 0.0), ('scale', 1.0))),
    'triangular': (_capi.SCG_RAND_TRIANGULAR, (('left', None), ('mode', None), ('right', None))),
    'randint': (_capi.SCG_RAND_RANDINT, (('low', None), ('high', None))),
    'rayleigh': (_capi.SCG_RAND_RAYLEIGH, (('scale', 1.0),)),
    'gumbel': (_capi.SCG_RAND_GUMBEL, (('loc', 0.0), ('scale', 1.0))),
    'weibull': (_capi.SCG_RAND_WEIBULL, (('a', None),)),
    'pareto': (_capi.SCG_RAND_PARETO, (('a', None),)),
    'power': (_capi.SCG_RAND_POWER, (('a', None),)),
    'lognormal': (_capi.SCG_RAND_LOGNORMAL, (('mean', 0.0), ('sigma', 1.0))),
    'standard_cauchy': (_capi.SCG_RAND_CAUCHY, ()),
    # aliases of kinds above
    'standard_exponential': (_capi.SCG_RAND_EXPONENTIAL, ()),
    'random_sample': (_capi.SCG_RAND_UNIFORM, ()), 'random': (_capi.SCG_RAND_UNIFORM, ()), 'ranf': (_capi.SCG_RAND_UNIFORM, ()),
    'sample': (_capi.SCG_RAND_UNIFORM, ()), 'rand': (_capi.SCG_RAND_UNIFORM, ()),
    'choice': (_capi.SCG_RAND_CHOICE, (('a', None), ('size', None), ('replace', True), ('p', None))),
    # rejection samplers (Marsaglia-Tsang gamma; beta from two gammas)
    'gamma': (_capi.SCG_RAND_GAMMA, (('shape', None), ('scale', 1.0))),
    'standard_gamma': (_capi.SCG_RAND_GAMMA, (('shape', None),)),
    'chisquare': (_capi.SCG_RAND_GAMMA, (('df', None),)),
    'beta': (_capi.SCG_RAND_BETA, (('a', None), ('b', None))),
}
SUPPORTED = tuple(sorted(_SIGNATURES))


def parse_distribution(info, what=''):
    """``{'distrib': name, 'args': [...], **kwargs}`` -> ``(kind, low, high, (p0, p1, p2))``.
    ``low`` / ``high`` are meaningful for ``uniform`` only; the other kinds read ``p``."""
    info = dict(info)
    name = info.pop('distrib')
    args = list(info.pop('args', []))
    if name not in _SIGNATURES:
        raise NotImplementedError('[ERROR] distrib %r%s is not available on the device (supported: %s).'
                                  % (name, ' for ' + what if what else '', ', '.join(SUPPORTED)))
    kind, params = _SIGNATURES[name]
    if name == 'choice':
        return _parse_choice(args, info, what)
    info.pop('size', None)
    if len(args) > len(params):
        raise TypeError('[ERROR] %s() takes at most %d positional arguments (%d given)%s.'
                        % (name, len(params), len(args), ' for ' + what if what else ''))
    vals = {}
    for k, (pname, default) in enumerate(params):
        if k < len(args):
            vals[pname] = args[k]
            if pname in info:
                raise TypeError('[ERROR] %s() got multiple values for argument %r.' % (name, pname))
        elif pname in info:
            vals[pname] = info.pop(pname)
        else:
            vals[pname] = default
    if info:
        raise TypeError('[ERROR] %s() got unexpected keyword arguments %s.' % (name, sorted(info)))
    if name == 'randint' and vals['high'] is None:                       # randint(low) draws from [0, low)
        vals['low'], vals['high'] = 0.0, vals['low']
    if any(v is None for v in vals.values()):
        raise TypeError('[ERROR] %s() is missing a required argument%s.' % (name, ' for ' + what if what else ''))
    v = [float(vals[p]) for p, _ in params]
    if name == 'standard_normal':
        v = [0.0, 1.0]
    if name == 'standard_exponential':
        v = [1.0]
    if name == 'standard_gamma':
        v = [v[0], 1.0]
    if name == 'chisquare':                                              # chisquare(df) = gamma(df / 2, scale 2)
        if not v[0] > 0:
            raise ValueError('[ERROR] df <= 0 in chisquare().')
        v = [0.5 * v[0], 2.0]
    if kind == _capi.SCG_RAND_GAMMA and (not v[0] > 0 or v[1] < 0):
        raise ValueError('[ERROR] %s() needs shape > 0 and scale >= 0%s.' % (name, ' for ' + what if what else ''))
    if kind == _capi.SCG_RAND_BETA and not (v[0] > 0 and v[1] > 0):
        raise ValueError('[ERROR] beta() needs a > 0 and b > 0%s.' % (' for ' + what if what else ''))
    if kind == _capi.SCG_RAND_UNIFORM:
        return (kind, v[0], v[1], (0.0, 0.0, 0.0)) if v else (kind, 0.0, 1.0, (0.0, 0.0, 0.0))
    if kind in (_capi.SCG_RAND_RAYLEIGH,) and v[0] < 0 or kind in (_capi.SCG_RAND_GUMBEL, _capi.SCG_RAND_LOGNORMAL) and v[1] < 0:
        raise ValueError('[ERROR] scale < 0 in %s()%s.' % (name, ' for ' + what if what else ''))
    if kind in (_capi.SCG_RAND_WEIBULL, _capi.SCG_RAND_PARETO, _capi.SCG_RAND_POWER) and not v[0] > 0:
        raise ValueError('[ERROR] a <= 0 in %s()%s.' % (name, ' for ' + what if what else ''))
    if kind in (_capi.SCG_RAND_NORMAL, _capi.SCG_RAND_LAPLACE, _capi.SCG_RAND_LOGISTIC) and v[1] < 0:
        raise ValueError('[ERROR] scale < 0 in %s()%s.' % (name, ' for ' + what if what else ''))
    if kind == _capi.SCG_RAND_EXPONENTIAL and v[0] < 0:
        raise ValueError('[ERROR] scale < 0 in exponential().')
    if kind == _capi.SCG_RAND_TRIANGULAR and not (v[0] <= v[1] <= v[2] and v[0] < v[2]):
        raise ValueError('[ERROR] triangular() needs left <= mode <= right and left < right.')
    if kind == _capi.SCG_RAND_RANDINT and not v[0] < v[1]:
        raise ValueError('[ERROR] randint() needs low < high.')
    return kind, 0.0, 0.0, tuple(v + [0.0] * (3 - len(v)))


def _parse_choice(args, info, what):
    """RandomState.choice(a, size=None, replace=True, p=None) for a scalar draw: `a` an int (= randint(0, a)) or a sequence of at
    most three values, equiprobable."""
    names = ('a', 'size', 'replace', 'p')
    if len(args) > len(names):
        raise TypeError('[ERROR] choice() takes at most 4 positional arguments.')
    vals = dict(zip(names, args))
    for k in list(info):
        if k in names:
            if k in vals:
                raise TypeError('[ERROR] choice() got multiple values for argument %r.' % k)
            vals[k] = info.pop(k)
    if info:
        raise TypeError('[ERROR] choice() got unexpected keyword arguments %s.' % sorted(info))
    if 'a' not in vals:
        raise TypeError('[ERROR] choice() is missing a required argument%s.' % (' for ' + what if what else ''))
    if vals.get('size') is not None:
        raise NotImplementedError('[ERROR] choice(size=...) draws an array; a randomised quantity is a scalar.')
    a, prob = vals['a'], vals.get('p')
    if isinstance(a, (int, float)) and not isinstance(a, bool):
        if int(a) != a or a < 1:
            raise ValueError('[ERROR] choice(a): a must be a positive integer or a sequence.')
        if prob is None:
            return _capi.SCG_RAND_RANDINT, 0.0, 0.0, (0.0, float(a), 0.0)
        a = list(range(int(a)))                                          # choice(3, p=...) draws from arange(3)
    a = [float(v) for v in a]
    if not 1 <= len(a) <= 3:
        raise NotImplementedError('[ERROR] choice() over %d values is not available on the device (at most 3).' % len(a))
    if prob is None:
        return _capi.SCG_RAND_CHOICE, float(len(a)), 0.0, tuple(a + [0.0] * (3 - len(a)))
    prob = [float(q) for q in prob]
    if len(prob) != len(a):
        raise ValueError("[ERROR] choice(): 'a' and 'p' must have same size")                     # numpy's own messages
    if any(q < 0 for q in prob):
        raise ValueError('[ERROR] choice(): probabilities are not non-negative')
    if abs(sum(prob) - 1.0) > 1e-8:
        raise ValueError('[ERROR] choice(): probabilities do not sum to 1')
    # weighted choice: low = P(a0), high = P(a0) + P(a1) (cumulative); unused value slots repeat the last value
    c1 = prob[0]
    c2 = 1.0 if len(a) <= 2 else prob[0] + prob[1]
    if len(a) == 1:
        c1 = 1.0
    return _capi.SCG_RAND_CHOICE_P, c1, c2, tuple(a + [a[-1]] * (3 - len(a)))


def lower_bound(kind, low, p):
    """The smallest value the draw can take (-inf when unbounded): used for the reference's positivity check of
    randomised inertial properties (quadrotor.py:464-465)."""
    if kind == _capi.SCG_RAND_UNIFORM:
        return low
    if kind == _capi.SCG_RAND_EXPONENTIAL:
        return 0.0
    if kind in (_capi.SCG_RAND_TRIANGULAR, _capi.SCG_RAND_RANDINT):
        return p[0]
    if kind in (_capi.SCG_RAND_RAYLEIGH, _capi.SCG_RAND_WEIBULL, _capi.SCG_RAND_PARETO, _capi.SCG_RAND_POWER, _capi.SCG_RAND_LOGNORMAL):
        return 0.0
    if kind == _capi.SCG_RAND_CHOICE:
        return min(p[:int(low)])
    if kind in (_capi.SCG_RAND_GAMMA, _capi.SCG_RAND_BETA):
        return 0.0
    if kind == _capi.SCG_RAND_CHOICE_P:
        return min(p)                                                    # (value slots without a probability repeat a real value)
    return float('-inf')
