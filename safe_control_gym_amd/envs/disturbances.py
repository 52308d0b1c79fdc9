"""Batched disturbances -- host-side mirror of the reference's ``envs/disturbances.py``.

Same class names, yaml keys (``disturbance_func: impulse | step | uniform | white_noise |
periodic``) and ``apply(target, env)`` protocol as /root/reference/safe_control_gym/envs/disturbances.py
(``ImpulseDisturbance`` :89-143, ``StepDisturbance`` :146-184, ``UniformNoise`` :186-212,
``WhiteNoise`` :215-248, ``PeriodicNoise`` :260-288, factory :315-333), but ``target`` is a
``[N, dim]`` device tensor, the step counter is per-env (``env.ctrl_step_counter`` is ``[N]``) and the
noise comes from a ``torch.Generator`` on the device seeded from the env seed (the reference shares
``env.np_random``, :39-43).  The resulting planes are handed to the fused step kernel, which adds
them exactly where the reference does (quadrotor.py:835-838 action, :526-548 force, :879-880 obs).
"""
import numpy as np
import torch


class Disturbance:
    def __init__(self, env, dim, mask=None, **kwargs):
        self.dim = dim
        self.mask = None
        if mask is not None:
            self.mask = torch.as_tensor(np.asarray(mask), dtype=torch.float32, device=env.device)
            assert self.dim == len(self.mask)
        self.generator = None

    def reset(self, env):
        pass

    def apply(self, target, env):
        return target

    def seed(self, env):
        self.generator = torch.Generator(device=env.device)
        self.generator.manual_seed(int(env._seed) % (2 ** 63))

    def _gen(self, env):
        if self.generator is None:
            self.seed(env)
        return self.generator

    def _finish(self, target, noise):
        if self.mask is not None:
            noise = noise * self.mask
        return target + noise


class DisturbanceList:
    def __init__(self, disturbances):
        self.disturbances = disturbances

    def reset(self, env):
        for d in self.disturbances:
            d.reset(env)

    def apply(self, target, env):
        out = target
        for d in self.disturbances:
            out = d.apply(out, env)
        return out

    def seed(self, env):
        for d in self.disturbances:
            d.seed(env)


class ImpulseDisturbance(Disturbance):
    def __init__(self, env, dim, mask=None, magnitude=1, step_offset=None, duration=1, decay_rate=1, **kwargs):
        super().__init__(env, dim, mask)
        self.magnitude = magnitude
        self.step_offset = step_offset
        self.max_step = int(env.EPISODE_LEN_SEC / env.CTRL_TIMESTEP)
        assert duration >= 1
        assert decay_rate > 0 and decay_rate <= 1
        self.duration = duration
        self.decay_rate = decay_rate

    def reset(self, env):
        n = env.num_envs
        if self.step_offset is None:
            self.current_step_offset = torch.randint(self.max_step, (n,), generator=self._gen(env), device=env.device)
        else:
            self.current_step_offset = torch.full((n,), int(self.step_offset), device=env.device)
        self.current_peak_step = (self.current_step_offset + self.duration / 2).to(torch.int64)

    def apply(self, target, env):
        t = env.ctrl_step_counter.to(torch.int64)
        peak_offset = (t - self.current_peak_step).abs().to(torch.float32)
        decay = torch.where(peak_offset < self.duration / 2, self.decay_rate ** peak_offset, torch.zeros_like(peak_offset))
        noise = torch.where(t >= self.current_step_offset, self.magnitude * decay, torch.zeros_like(decay))
        return self._finish(target, noise[:, None].expand_as(target))


class StepDisturbance(Disturbance):
    def __init__(self, env, dim, mask=None, magnitude=1, step_offset=None, **kwargs):
        super().__init__(env, dim, mask)
        self.magnitude = magnitude
        self.step_offset = step_offset
        self.max_step = int(env.EPISODE_LEN_SEC / env.CTRL_TIMESTEP)

    def reset(self, env):
        n = env.num_envs
        if self.step_offset is None:
            self.current_step_offset = torch.randint(self.max_step, (n,), generator=self._gen(env), device=env.device)
        else:
            self.current_step_offset = torch.full((n,), int(self.step_offset), device=env.device)

    def apply(self, target, env):
        t = env.ctrl_step_counter.to(torch.int64)
        noise = torch.where(t >= self.current_step_offset, float(self.magnitude), 0.0)
        return self._finish(target, noise[:, None].expand_as(target))


class UniformNoise(Disturbance):
    def __init__(self, env, dim, mask=None, low=0.0, high=1.0, **kwargs):
        super().__init__(env, dim, mask)
        if isinstance(low, float):
            low = [low] * self.dim
        elif not isinstance(low, list):
            raise ValueError('[ERROR] UniformNoise.__init__(): low must be specified as a float or list.')
        if isinstance(high, float):
            high = [high] * self.dim
        elif not isinstance(high, list):
            raise ValueError('[ERROR] UniformNoise.__init__(): high must be specified as a float or list.')
        self.low = torch.as_tensor(low, dtype=torch.float32, device=env.device)
        self.high = torch.as_tensor(high, dtype=torch.float32, device=env.device)

    def apply(self, target, env):
        u = torch.rand(target.shape, generator=self._gen(env), device=target.device, dtype=torch.float32)
        return self._finish(target, self.low + (self.high - self.low) * u)


class WhiteNoise(Disturbance):
    def __init__(self, env, dim, mask=None, std=1.0, **kwargs):
        super().__init__(env, dim, mask)
        if isinstance(std, float):
            std = [std] * self.dim
        elif not isinstance(std, list):
            raise ValueError('[ERROR] WhiteNoise.__init__(): std must be specified as a float or list.')
        assert self.dim == len(std), 'std shape should be the same as dim.'
        self.std = torch.as_tensor(std, dtype=torch.float32, device=env.device)

    def apply(self, target, env):
        g = torch.randn(target.shape, generator=self._gen(env), device=target.device, dtype=torch.float32)
        return self._finish(target, g * self.std)


class PeriodicNoise(Disturbance):
    def __init__(self, env, dim, mask=None, scale=1.0, frequency=1.0, **kwargs):
        super().__init__(env, dim)
        self.scale = scale
        self.frequency = frequency

    def apply(self, target, env):
        phase = (torch.rand(target.shape, generator=self._gen(env), device=target.device) * 2 - 1) * np.pi
        t = env.pyb_step_counter.to(torch.float32)[:, None] * env.PYB_TIMESTEP
        return self._finish(target, self.scale * torch.sin(2 * np.pi * self.frequency * t + phase))


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
