"""Disturbances -- host-side mirror of the reference's ``envs/disturbances.py``, compiled for the step kernel.

Same class names, yaml keys (``disturbance_func: impulse | step | uniform | white_noise | periodic``),
constructor arguments, defaults and assertions as /root/reference/safe_control_gym/envs/disturbances.py
(``ImpulseDisturbance`` :89-143, ``StepDisturbance`` :146-184, ``UniformNoise`` :186-212, ``WhiteNoise``
:215-248, ``PeriodicNoise`` :260-288, factory :315-333).  The reference evaluates ``apply(target, env)`` in
NumPy once per env and step; here every object only validates and holds its parameters and ``compile()``s into
one ``scg_disturbance`` entry of the C ABI (include/scg_rollout.h): the noise itself is generated and added
inside the fused HIP step kernel (``apply_disturbances`` in csrc/scg_kernels.hip), exactly where the reference
adds it (quadrotor.py:835-836 action, :526-548 force, :879-880 observation), from per-env Philox streams
instead of the shared ``env.np_random`` (:39-43).
"""
import numpy as np

from safe_control_gym_amd import _capi

_MODES = {'action': _capi.SCG_DIST_ACTION, 'dynamics': _capi.SCG_DIST_DYNAMICS,
          'observation': _capi.SCG_DIST_OBSERVATION}


class Disturbance:
    KIND = None

    def __init__(self, env, dim, mask=None, **kwargs):
        self.dim = dim
        self.mask = mask
        if mask is not None:
            self.mask = np.asarray(mask)
            assert self.dim == len(self.mask)

    def reset(self, env):
        """Per-episode quantities (the random ``step_offset``) are derived in the kernel from the episode id."""

    def seed(self, env):
        """The kernel's Philox streams are keyed by the env seed (scg_config.seed); nothing to do here."""

    def _entry(self, mode):
        d = _capi.ScgDisturbance()
        d.mode, d.kind = _MODES[mode], self.KIND
        d.step_offset, d.max_step, d.duration = 0, 0, 1
        d.decay_rate = 1.0
        if self.mask is not None:
            d.has_mask = 1
            for j in range(self.dim):
                d.mask[j] = float(self.mask[j])
        return d

    def compile(self, mode):
        raise NotImplementedError


class DisturbanceList:
    def __init__(self, disturbances):
        self.disturbances = disturbances

    def reset(self, env):
        for d in self.disturbances:
            d.reset(env)

    def seed(self, env):
        for d in self.disturbances:
            d.seed(env)

    def compile(self, mode):
        assert len(self.disturbances) <= _capi.SCG_MAX_DISTURBANCES, \
            '[ERROR] at most %d disturbances per mode.' % _capi.SCG_MAX_DISTURBANCES
        return [d.compile(mode) for d in self.disturbances]

    @property
    def needs_episode_id(self):
        return any(getattr(d, 'step_offset', 0) is None for d in self.disturbances)


class ImpulseDisturbance(Disturbance):
    KIND = _capi.SCG_DIST_IMPULSE

    def __init__(self, env, dim, mask=None, magnitude=1, step_offset=None, duration=1, decay_rate=1, **kwargs):
        super().__init__(env, dim, mask)
        self.magnitude = magnitude
        self.step_offset = step_offset
        self.max_step = int(env.EPISODE_LEN_SEC / env.CTRL_TIMESTEP)
        assert duration >= 1
        assert decay_rate > 0 and decay_rate <= 1
        self.duration = duration
        self.decay_rate = decay_rate

    def compile(self, mode):
        d = self._entry(mode)
        d.magnitude, d.decay_rate, d.duration = float(self.magnitude), float(self.decay_rate), int(self.duration)
        d.step_offset = -1 if self.step_offset is None else int(self.step_offset)
        d.max_step = self.max_step
        return d


class StepDisturbance(Disturbance):
    KIND = _capi.SCG_DIST_STEP

    def __init__(self, env, dim, mask=None, magnitude=1, step_offset=None, **kwargs):
        super().__init__(env, dim, mask)
        self.magnitude = magnitude
        self.step_offset = step_offset
        self.max_step = int(env.EPISODE_LEN_SEC / env.CTRL_TIMESTEP)

    def compile(self, mode):
        d = self._entry(mode)
        d.magnitude = float(self.magnitude)
        d.step_offset = -1 if self.step_offset is None else int(self.step_offset)
        d.max_step = self.max_step
        return d


class UniformNoise(Disturbance):
    KIND = _capi.SCG_DIST_UNIFORM

    def __init__(self, env, dim, mask=None, low=0.0, high=1.0, **kwargs):
        super().__init__(env, dim, mask)
        if isinstance(low, float):
            self.low = np.asarray([low] * self.dim)
        elif isinstance(low, list):
            self.low = np.asarray(low)
        else:
            raise ValueError('[ERROR] UniformNoise.__init__(): low must be specified as a float or list.')
        if isinstance(high, float):
            self.high = np.asarray([high] * self.dim)
        elif isinstance(high, list):
            self.high = np.asarray(high)
        else:
            raise ValueError('[ERROR] UniformNoise.__init__(): high must be specified as a float or list.')
        assert len(self.low) == self.dim and len(self.high) == self.dim, 'low / high shape should be the same as dim.'

    def compile(self, mode):
        d = self._entry(mode)
        for j in range(self.dim):
            d.low[j], d.high[j] = float(self.low[j]), float(self.high[j])
        return d


class WhiteNoise(Disturbance):
    KIND = _capi.SCG_DIST_WHITE_NOISE

    def __init__(self, env, dim, mask=None, std=1.0, **kwargs):
        super().__init__(env, dim, mask)
        if isinstance(std, float):
            self.std = np.asarray([std] * self.dim)
        elif isinstance(std, list):
            self.std = np.asarray(std)
        else:
            raise ValueError('[ERROR] WhiteNoise.__init__(): std must be specified as a float or list.')
        assert self.dim == len(self.std), 'std shape should be the same as dim.'

    def compile(self, mode):
        d = self._entry(mode)
        for j in range(self.dim):
            d.low[j] = float(self.std[j])
        return d


class PeriodicNoise(Disturbance):
    KIND = _capi.SCG_DIST_PERIODIC

    def __init__(self, env, dim, mask=None, scale=1.0, frequency=1.0, **kwargs):
        super().__init__(env, dim)               # the reference drops `mask` here too (disturbances.py:273)
        self.scale = scale
        self.frequency = frequency

    def compile(self, mode):
        d = self._entry(mode)
        d.scale, d.frequency = float(self.scale), float(self.frequency)
        return d


DISTURBANCE_TYPES = {'impulse': ImpulseDisturbance, 'step': StepDisturbance, 'uniform': UniformNoise,
                     'white_noise': WhiteNoise, 'periodic': PeriodicNoise}


def create_disturbance_list(disturbance_specs, shared_args, env):
    out = []
    for spec in disturbance_specs:
        assert 'disturbance_func' in spec.keys(), '[ERROR]: Every distrubance must specify a disturbance_func.'
        func = spec['disturbance_func']
        assert func in DISTURBANCE_TYPES, '[ERROR] in BenchmarkEnv._setup_disturbances(), disturbance type not available.'
        cfg = {k: v for k, v in spec.items() if k != 'disturbance_func'}
        out.append(DISTURBANCE_TYPES[func](env, **shared_args, **cfg))
    return DisturbanceList(out)
