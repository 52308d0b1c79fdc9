"""Oracle: the reference's disturbance classes restated in float64, driven by Philox draws.

Test infrastructure -- see ``oracle/__init__.py``.

Follows /root/reference/safe_control_gym/envs/disturbances.py: ``ImpulseDisturbance`` (:89-143),
``StepDisturbance`` (:146-184), ``UniformNoise`` (:186-212), ``WhiteNoise`` (:215-248), ``PeriodicNoise``
(:260-288, which drops its ``mask`` argument, :273), ``DisturbanceList.apply`` (:66-74: entries applied in list
order), and the wiring of the three modes in quadrotor.py:526-548 (dynamics: 1-D -> z, 2-D -> (x, z)),
:835-836 (action) and :879-880 (observation).  ``ctrl_step_counter`` is read BEFORE its increment
(benchmark_env.py:427-428 runs in after_step).

The reference draws from ``env.np_random`` (MT19937).  The device engine cannot reproduce that stream (DESIGN.md
section 3), so both sides take their uniforms from the Philox streams documented in
``safe_control_gym_amd/csrc/scg_kernels.hip`` (``apply_disturbances``): normal draws are Box-Muller pairs of those
uniforms.  Everything AFTER the raw draw is pinned against the live reference classes in
tests/test_disturbances.py (the reference module imports without casadi / pybullet / gym).
"""
import math

import numpy as np

from oracle import philox

MODES = {'action': 0, 'dynamics': 1, 'observation': 2}
MAX_DISTURBANCES = 4
CTR_BASE, EPISODE_CTR = 16, 128
TWO_M24 = 2.0 ** -24


def _uniforms(seed, env_index, serial, word3_base, dim):
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    out = []
    for b in range((dim + 3) // 4):
        ctr = np.array([env_index & 0xFFFFFFFF, (env_index >> 32) & 0xFFFFFFFF, serial & 0xFFFFFFFF,
                        word3_base + b], dtype=np.uint32)
        out.extend(philox.uniform01(philox.philox4x32_10(ctr, key)))
    return np.asarray(out, dtype=np.float64)


def episode_step_offset(seed, env_index, episode, mode, entry, max_step):
    """np_random.randint(max_step) (:128-131) from Philox word 0: floor(word * max_step / 2^32)."""
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    ctr = np.array([env_index & 0xFFFFFFFF, (env_index >> 32) & 0xFFFFFFFF, episode & 0xFFFFFFFF,
                    EPISODE_CTR + mode * MAX_DISTURBANCES + entry], dtype=np.uint32)
    w0 = int(philox.philox4x32_10(ctr, key)[0])
    return (w0 * int(max_step)) >> 32


def standard_normals(u):
    """Box-Muller on consecutive pairs (u1 + 2^-24 in (0, 1], u2): z0 = r cos(2 pi u2), z1 = r sin(2 pi u2)."""
    z = np.zeros(len(u))
    for k in range(len(u) // 2):
        r = math.sqrt(-2.0 * math.log(u[2 * k] + TWO_M24))
        z[2 * k] = r * math.cos(2.0 * math.pi * u[2 * k + 1])
        z[2 * k + 1] = r * math.sin(2.0 * math.pi * u[2 * k + 1])
    return z


def entry_noise(spec, dim, t, step_offset, u, ctrl_timestep):
    """Noise vector (dim,) of one entry at pre-increment step t.  ``u``: this entry's uniforms (>= dim, even count);
    ``step_offset``: the episode's offset for impulse / step entries."""
    func = spec['disturbance_func']
    mask = spec.get('mask')
    if func == 'impulse':                                            # :126-142
        duration, decay_rate, magnitude = spec.get('duration', 1), spec.get('decay_rate', 1), spec.get('magnitude', 1)
        noise = 0.0
        if t >= step_offset:
            peak = int(step_offset + duration / 2)
            peak_offset = abs(t - peak)
            noise = magnitude * (decay_rate ** peak_offset if peak_offset < duration / 2 else 0.0)
        noise = np.full(dim, float(noise))
    elif func == 'step':                                             # :173-183
        noise = np.full(dim, float(spec.get('magnitude', 1)) if t >= step_offset else 0.0)
    elif func == 'uniform':                                          # :207-211
        low, high = _vec(spec.get('low', 0.0), dim), _vec(spec.get('high', 1.0), dim)
        noise = low + (high - low) * u[:dim]
    elif func == 'white_noise':                                      # :241-247
        noise = standard_normals(u)[:dim] * _vec(spec.get('std', 1.0), dim)
    elif func == 'periodic':                                         # :282-287 (no mask: :273)
        phase = -math.pi + 2.0 * math.pi * u[:dim]
        noise = spec.get('scale', 1.0) * np.sin(2.0 * math.pi * spec.get('frequency', 1.0) * (t * ctrl_timestep) + phase)
        mask = None
    else:
        raise ValueError(func)
    if mask is not None:
        noise = noise * np.asarray(mask, dtype=np.float64)
    return noise


def _vec(v, dim):
    return np.full(dim, float(v)) if isinstance(v, (int, float)) else np.asarray(v, dtype=np.float64)


class RefDisturbanceModel:
    """Per-env driver: fills RefEnv.action_disturbance / dynamics_force / obs_noise before each step and the
    observation noise of each reset.  ``serial`` mirrors the engine's step serial and is set by the test harness."""

    def __init__(self, specs, system, seed, env_index, episode_len_sec, ctrl_timestep):
        self.specs = {m: list(specs.get(m, [])) for m in MODES}
        self.system, self.seed, self.env_index = system, seed, env_index
        self.max_step = int(episode_len_sec / ctrl_timestep)
        self.dt = ctrl_timestep
        nx = {1: 2, 2: 6, 3: 12, 'cartpole': 4}[system]
        nu = {1: 1, 2: 2, 3: 4, 'cartpole': 1}[system]
        self.dims = {'action': nu, 'dynamics': {1: 1, 2: 2, 3: 3, 'cartpole': 1}[system], 'observation': nx}
        self.serial = 0
        self.episode = 0

    def _mode_noise(self, mode, stream, t, serial, episode):
        dim = self.dims[mode]
        total = np.zeros(dim)
        m = MODES[mode]
        for e, spec in enumerate(self.specs[mode]):
            base = CTR_BASE + ((stream * 3 + m) * MAX_DISTURBANCES + e) * 4
            u = _uniforms(self.seed, self.env_index, serial, base, dim)
            off = spec.get('step_offset')
            if off is None and spec['disturbance_func'] in ('impulse', 'step'):
                off = episode_step_offset(self.seed, self.env_index, episode, m, e, self.max_step)
            total = total + entry_noise(spec, dim, t, off, u, self.dt)
        return total

    def on_reset(self, env):
        self.episode = self.serial
        env.obs_noise = self._mode_noise('observation', 1, 0, self.serial, self.serial) if self.specs['observation'] else None

    def on_step(self, env):
        t = env.ctrl_step_counter
        env.action_disturbance = self._mode_noise('action', 0, t, self.serial, self.episode) if self.specs['action'] else None
        if self.specs['dynamics']:
            d = self._mode_noise('dynamics', 0, t, self.serial, self.episode)
            if self.system == 1:
                env.dynamics_force = np.array([0.0, 0.0, d[0]])                 # quadrotor.py:540-542
            elif self.system == 2:
                env.dynamics_force = np.array([d[0], 0.0, d[1]])                # :543-545
            elif self.system == 3:
                env.dynamics_force = d
            else:
                env.dynamics_force = np.array([d[0], 0.0, 0.0])                 # cart force
        env.obs_noise = self._mode_noise('observation', 0, t, self.serial, self.episode) if self.specs['observation'] else None
