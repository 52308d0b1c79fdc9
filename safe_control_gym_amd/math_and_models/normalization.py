"""Normalisers for observations and rewards -- device mirror of the reference's ``math_and_models/normalization.py``.

Same classes and call protocol as /root/reference/safe_control_gym/math_and_models/normalization.py
(``RunningMeanStd`` :17-62, ``BaseNormalizer`` :65-104, ``MeanStdNormalizer`` :107-147, ``RewardStdNormalizer``
:150-190, ``RescaleNormalizer`` :193-214, ``ActionUnnormalizer`` :226-248, ``normalize_angle`` :10-14), but the
batch is a device tensor straight out of the step kernel and the tracking + normalisation run in two HIP
kernels behind ``scg_mean_std_normalize`` / ``scg_reward_normalize`` (csrc/scg_normalize.hip).  The running
statistics live in float64 device records (mean[d], var[d], count), double-buffered so that an update never
needs a host round trip; ``state_dict()`` / ``load_state_dict()`` exchange the reference's ``{'mean', 'var'}``
NumPy arrays.

Observations are taken zero-copy when they are the engine's own ``[N, d]`` view of SoA planes
(``env.step()``'s obs: stride (1, ld)); any other tensor is transposed into planes first.
"""
import functools
import math
import weakref

import numpy as np
import torch

from safe_control_gym_amd import _capi
from safe_control_gym_amd.utils.recycle import RecyclingPool


def normalize_angle(x):
    """Wraps input angle to [-pi, pi] (normalization.py:10-14)."""
    return ((x + math.pi) % (2 * math.pi)) - math.pi


_LIVE_TRACKERS = weakref.WeakSet()


def live_trackers():
    """Every RunningMeanStd alive in this process (envs/graph_loop.py checks that a recorded stretch leaves each one's double-buffered
    record where it found it)."""
    return list(_LIVE_TRACKERS)


class RunningMeanStd:
    """Device-resident running mean / variance / count (normalization.py:17-62).  ``update`` is fused into the
    normalisers' kernels; this object owns the two float64 state records and exposes the current one."""

    def __init__(self, epsilon=1e-4, shape=(), device='cuda'):
        self.shape = tuple(shape) if not isinstance(shape, int) else (shape,)
        self.d = int(np.prod(self.shape)) if self.shape else 1
        self.device = torch.device(device)
        self._state = torch.zeros((2, 2 * self.d + 1), dtype=torch.float64, device=self.device)
        self._state[:, self.d:2 * self.d] = 1.0
        self._state[:, 2 * self.d] = epsilon
        self._cur = 0
        self._moments = torch.zeros(max(2 * max(64 * self.d, 1024), 3072), dtype=torch.float64, device=self.device)   # kernel scratch
        # raw addresses for the per-step calls (the records never move): a tensor index + data_ptr() costs ~3 us of host each
        self._rec_ptr = (self._state[0].data_ptr(), self._state[1].data_ptr())
        self._moments_ptr = self._moments.data_ptr()
        self.device = self._state.device                        # with its index: the launches check the current device against it
        _LIVE_TRACKERS.add(self)

    # the reference's attributes, as views of the current record
    @property
    def mean(self):
        return self._state[self._cur, :self.d].reshape(self.shape)

    @property
    def var(self):
        return self._state[self._cur, self.d:2 * self.d].reshape(self.shape)

    @property
    def count(self):
        return self._state[self._cur, 2 * self.d]

    def _records(self, update):
        """(state_in, state_out or None); flips the current record when updating."""
        cur = self._state[self._cur]
        if not update:
            return cur, None
        nxt = self._state[1 - self._cur]
        self._cur = 1 - self._cur
        return cur, nxt

    def _record_ptrs(self, update):
        """``_records`` as raw addresses (state_in, state_out or None)."""
        cur = self._rec_ptr[self._cur]
        if not update:
            return cur, None
        self._cur = 1 - self._cur
        return cur, self._rec_ptr[self._cur]

    def update(self, arr):
        """normalization.py:34-46 on a device batch [N, *shape]: the moments pass of the normalisers alone (``y == NULL``:
        no scratch output, no apply pass).  An [N, d] view of SoA planes (``env.step()``'s obs) is read in place."""
        x = torch.as_tensor(arr, device=self.device)
        if x.dtype != torch.float32:
            x = x.to(torch.float32)
        x = x.reshape(-1, self.d)
        n = x.shape[0]
        if n == 0:
            return
        if x.stride(0) == 1 and (self.d == 1 or x.stride(1) >= n):
            planes, ld = x, (x.stride(1) if self.d > 1 else n)             # already [d][ld] planes underneath: zero copy
        else:
            planes, ld = x.T.contiguous(), n                                # [d, n] SoA planes
        state_in, state_out = self._record_ptrs(True)
        _launch(self.device, _capi.load().scg_mean_std_normalize, 'scg_mean_std_normalize', planes.data_ptr(), 0, self.d, ld, 0, n,
                state_in, state_out, self._moments_ptr, 0.0, 0.0)

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        """normalization.py:48-62 (Chan's parallel update), float64 on the device record; device tensors stay on the device."""
        def as64(v):
            if isinstance(v, torch.Tensor):
                return v.detach().to(device=self.device, dtype=torch.float64).reshape(-1)
            return torch.as_tensor(np.asarray(v, dtype=np.float64).reshape(-1), device=self.device)
        bm, bv = as64(batch_mean), as64(batch_var)
        bc = batch_count.detach().to(device=self.device, dtype=torch.float64) if isinstance(batch_count, torch.Tensor) else float(batch_count)
        cur, nxt = self._records(True)
        mean, var, count = cur[:self.d], cur[self.d:2 * self.d], cur[2 * self.d]
        delta = bm - mean
        tot = count + bc
        nxt[:self.d] = mean + delta * bc / tot
        nxt[self.d:2 * self.d] = (var * count + bv * bc + delta * delta * count * bc / tot) / tot
        nxt[2 * self.d] = tot

    def set(self, mean=None, var=None, count=None):
        rec = self._state[self._cur]
        if mean is not None:
            rec[:self.d] = torch.as_tensor(np.asarray(mean, dtype=np.float64).reshape(-1), device=self.device)
        if var is not None:
            rec[self.d:2 * self.d] = torch.as_tensor(np.asarray(var, dtype=np.float64).reshape(-1), device=self.device)
        if count is not None:
            rec[2 * self.d] = float(count)


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _launch(device, fn, name, *args):
    """``fn(*args, current stream of device)`` with the device current; without the context manager (~5 us) when it already is."""
    idx = device.index
    if _RAW_STREAM is not None and idx is not None and torch.cuda.current_device() == idx:
        status = fn(*args, _RAW_STREAM(idx))
    else:
        with torch.cuda.device(device):
            status = fn(*args, _capi.current_stream(device))
    if status != 0:
        _capi.check(status, name)


def _new_plane_set(d, ld, n, device):
    """A normalised-observation set for the recycling pool: [d, ld] planes and the [n, d] view of them that is returned."""
    out = torch.empty((d, ld), dtype=torch.float32, device=device)
    view = out.T if ld == n else out[:, :n].T
    return (out, view), (out, view), out.untyped_storage()


def _new_row_set(n, device):
    out = torch.empty(n, dtype=torch.float32, device=device)
    return out, (out,), out.untyped_storage()


class _FreshOutputs:
    """The fresh result tensor of a normaliser call, recycled once the caller holds no reference to the last ones (utils/recycle.py:
    the reference returns a new array per call, normalization.py:125-133, and its loops keep the previous one).  One pool per batch
    size (the last one seen); up to 8 sets within 64 MiB, else plain allocation."""

    __slots__ = ('key', 'pool')

    def __init__(self):
        self.key, self.pool = None, None

    def take(self, key, set_bytes, factory, *args):
        if key != self.key:
            cap = min(8, (64 << 20) // max(1, set_bytes))
            self.key, self.pool = key, (RecyclingPool(functools.partial(factory, *args), cap) if cap >= 4 else None)
        return None if self.pool is None else self.pool.take()


class BaseNormalizer:
    """Template / default normaliser: identity (normalization.py:65-104)."""

    def __init__(self, read_only=False):
        self.read_only = read_only

    def set_read_only(self):
        self.read_only = True

    def unset_read_only(self):
        self.read_only = False

    def __call__(self, x, *args, **kwargs):
        return x

    def state_dict(self):
        return {}

    def load_state_dict(self, _):
        pass


class _SharedStatistics:
    """One tracker, the batch spread over the ranks of a ``torch.distributed`` world (one process per GPU, each with a shard of the
    envs): the reference normalises in ONE process over the observations of ALL its envs (controllers/ppo/ppo.py:260-261), so the
    shards' moments are added up between the two passes of the normaliser -- one all-reduce of 2 d (+1) doubles per call, over RCCL
    / xGMI on the device (through host memory under gloo) -- and every rank ends up with the record a single tracker fed with the
    whole batch would hold.  Every rank must make the same calls with a fixed shard size (the total row count is exchanged once per
    shard size, the only blocking exchange)."""

    def __init__(self, sync_statistics, process_group):
        # 'force': the exchange also in a world of ONE rank (tests: the RCCL calls on a one-GPU box, where a second rank cannot live)
        self.sync_statistics, self.force, self.process_group = bool(sync_statistics), sync_statistics == 'force', process_group
        self._totals = {}

    def world(self):
        """torch.distributed when the statistics are shared and there is more than one rank, else None."""
        if not self.sync_statistics:
            return None
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and (self.force or dist.get_world_size(self.process_group) > 1):
            return dist
        return None

    def add_up(self, dist, t):
        """In-place sum of a small device tensor over the ranks."""
        if dist.get_backend(self.process_group) == 'gloo' and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h, group=self.process_group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.process_group)

    def total_rows(self, dist, n, device):
        if n not in self._totals:
            t = torch.tensor([float(n)], dtype=torch.float64, device=device)
            self.add_up(dist, t)
            self._totals[n] = float(t.item())
        return self._totals[n]


class MeanStdNormalizer(BaseNormalizer):
    """Normalise by the running average (normalization.py:107-147).  ``sync_statistics=True`` (with an initialised
    ``torch.distributed`` world of more than one rank): the tracker is updated with the batch of ALL ranks (``_SharedStatistics``)."""

    def __init__(self, shape=(), read_only=False, clip=10.0, epsilon=1e-8, device='cuda', sync_statistics=False, process_group=None):
        super().__init__(read_only)
        self.lib = _capi.load()
        self.rms = RunningMeanStd(shape=shape, device=device)
        self.clip = clip
        self.epsilon = epsilon
        self.shared = _SharedStatistics(sync_statistics, process_group)
        self._sums = None
        self.device = self.rms.device
        self._fresh = _FreshOutputs()

    def _planes(self, x):
        """(planes pointer tensor, ld, n): zero copy for an [N, d] view of SoA planes, else one transposing copy."""
        d = self.rms.d
        if type(x) is torch.Tensor and x.dtype == torch.float32 and x.ndim == 2 and x.device == self.device and x.shape[1] == d:
            n = x.shape[0]                                                  # the per-step case: env.step()'s obs
            if n > 0 and x.stride(0) == 1 and (d == 1 or x.stride(1) >= n):
                return x, (x.stride(1) if d > 1 else n), n
        x = torch.as_tensor(x, device=self.device)
        if x.dtype != torch.float32:
            x = x.to(torch.float32)
        if x.ndim == 1:
            x = x.reshape(-1, 1) if d == 1 else x.reshape(1, d)
        if x.ndim != 2 or x.shape[1] != d:
            raise ValueError('[ERROR] expected a (batch, %d) input, got %s' % (d, tuple(x.shape)))
        n = x.shape[0]
        if n > 0 and x.stride(0) == 1 and (d == 1 or x.stride(1) >= n):
            return x, (x.stride(1) if d > 1 else n), n
        t = x.T.contiguous()
        return t.T, n, n

    def __call__(self, x, out=None):
        """Update the tracker with the batch (unless read_only), return the normalised, clipped batch [N, d].

        The result is a FRESH tensor on every call, as the reference's is (normalization.py:125-133 returns a new
        array): callers keep the previous step's observation around while the next one is produced
        (controllers/ppo/ppo.py:259-276 pushes ``obs`` after ``obs_normalizer(next_obs)`` has run).  ``out`` opts into
        reuse: a float32 [d, ld] plane buffer (ld >= N, a multiple of 64) that receives the result and whose [N, d]
        view is returned -- then the caller owns the aliasing."""
        xv, ld_x, n = self._planes(x)
        d = self.rms.d
        if n == 0:
            if not self.read_only and self.shared.world() is not None:
                raise ValueError('[ERROR] sync_statistics: an empty shard cannot take part in the exchange (every rank makes every call).')
            return xv
        ld_y = (n + 63) // 64 * 64
        view = None
        if out is None:
            got = self._fresh.take((d, n), 4 * d * ld_y, _new_plane_set, d, ld_y, n, self.device)
            if got is not None:
                out, view = got
                got = None
            else:
                out = torch.empty((d, ld_y), dtype=torch.float32, device=self.device)
        elif not (isinstance(out, torch.Tensor) and out.dtype == torch.float32 and out.is_contiguous() and out.ndim == 2
                  and out.shape[0] == d and out.shape[1] >= n and out.device == xv.device):
            raise ValueError('[ERROR] out must be a contiguous float32 (%d, ld >= %d) device tensor' % (d, n))
        ld_y = out.shape[1]
        dist = None if self.read_only else self.shared.world()
        state_in, state_out = self.rms._record_ptrs(not self.read_only)
        if dist is not None:
            # moments of this shard -> sums over the ranks -> update from the added sums + normalise this shard
            if self._sums is None:
                self._sums = torch.zeros(2 * d, dtype=torch.float64, device=self.device)
            total = self.shared.total_rows(dist, n, self.device)
            _launch(self.device, self.lib.scg_mean_std_moments, 'scg_mean_std_moments', xv.data_ptr(), d, ld_x, n, state_in,
                    self.rms._moments_ptr, self._sums.data_ptr())
            self.shared.add_up(dist, self._sums)
            _launch(self.device, self.lib.scg_mean_std_normalize_sums, 'scg_mean_std_normalize_sums', xv.data_ptr(), out.data_ptr(), d,
                    ld_x, ld_y, n, state_in, state_out, self._sums.data_ptr(), total, float(self.clip), float(self.epsilon))
            return view if view is not None else out[:, :n].T
        _launch(self.device, self.lib.scg_mean_std_normalize, 'scg_mean_std_normalize', xv.data_ptr(), out.data_ptr(), d, ld_x, ld_y, n,
                state_in, state_out, self.rms._moments_ptr, float(self.clip), float(self.epsilon))
        return view if view is not None else out[:, :n].T

    def state_dict(self):
        return {'mean': self.rms.mean.cpu().numpy(), 'var': self.rms.var.cpu().numpy()}

    def load_state_dict(self, saved):
        self.rms.set(mean=saved['mean'], var=saved['var'])


class RewardStdNormalizer(MeanStdNormalizer):
    """Reward normalisation by the running std of the discounted returns (normalization.py:150-190).

    ``legacy_index_reset=True`` reproduces the reference's ``self.ret[dones.astype(np.long)] = 0`` literally (it
    clears ret[0] / ret[1] according to the VALUES in dones); the default clears the returns of the finished envs,
    which is what the comment above that line says it is for."""

    def __init__(self, gamma=0.99, read_only=False, clip=10.0, epsilon=1e-8, device='cuda', legacy_index_reset=False,
                 sync_statistics=False, process_group=None):
        super().__init__((), read_only, clip, epsilon, device=device, sync_statistics=sync_statistics, process_group=process_group)
        self.gamma = gamma
        self.ret = None
        self.legacy_index_reset = legacy_index_reset
        if sync_statistics and legacy_index_reset:
            raise ValueError('[ERROR] sync_statistics: the legacy index reset addresses elements 0 / 1 of ONE array of returns; the shards '
                             'of a distributed batch clear the returns of their finished envs (legacy_index_reset=False).')

    def __call__(self, x, dones, out=None):
        """``out`` (optional): a contiguous float32 [>= N] device tensor that receives the normalised rewards (e.g. row t of a
        rollout buffer's reward planes); its [N] head is returned."""
        dev = self.device
        if not (type(x) is torch.Tensor and x.dtype == torch.float32 and x.ndim == 1 and x.device == dev and x.is_contiguous()):
            x = torch.as_tensor(x, dtype=torch.float32, device=dev).reshape(-1).contiguous()
        n = x.shape[0]
        if n == 0:
            if not self.read_only and self.shared.world() is not None:
                raise ValueError('[ERROR] sync_statistics: an empty shard cannot take part in the exchange (every rank makes every call).')
            return x
        if not (type(dones) is torch.Tensor and dones.dtype in (torch.bool, torch.uint8) and dones.ndim == 1 and dones.device == dev
                and dones.is_contiguous()):                    # (the per-step case, env.step()'s rew / done, skips both conversions)
            dones = torch.as_tensor(dones, device=dev).reshape(-1)
            if dones.dtype not in (torch.bool, torch.uint8):
                dones = dones != 0
            dones = dones.contiguous()
        if not self.read_only and self.ret is None:
            self.ret = torch.zeros(n, dtype=torch.float32, device=self.device)
        if out is None:
            out = self._fresh.take(n, 4 * n, _new_row_set, n, self.device)
            if out is None:
                out = torch.empty(n, dtype=torch.float32, device=self.device)
        elif not (isinstance(out, torch.Tensor) and out.dtype == torch.float32 and out.ndim == 1 and out.is_contiguous()
                  and out.shape[0] >= n and out.device == x.device):
            raise ValueError('[ERROR] out must be a contiguous float32 (>= %d,) device tensor' % n)
        dist = None if self.read_only else self.shared.world()
        state_in, state_out = self.rms._record_ptrs(not self.read_only)
        if dist is not None:
            if self._sums is None:
                self._sums = torch.zeros(3, dtype=torch.float64, device=dev)
            total = self.shared.total_rows(dist, n, dev)
            _launch(dev, self.lib.scg_reward_moments, 'scg_reward_moments', x.data_ptr(), dones.data_ptr(), self.ret.data_ptr(), n, state_in,
                    self.rms._moments_ptr, self._sums.data_ptr(), float(self.gamma))
            self.shared.add_up(dist, self._sums)
            _launch(dev, self.lib.scg_reward_normalize_sums, 'scg_reward_normalize_sums', x.data_ptr(), out.data_ptr(), n, state_in, state_out,
                    self._sums.data_ptr(), total, float(self.clip), float(self.epsilon))
            return out if out.shape[0] == n else out[:n]
        _launch(dev, self.lib.scg_reward_normalize, 'scg_reward_normalize', x.data_ptr(), dones.data_ptr(),
                None if self.ret is None else self.ret.data_ptr(), out.data_ptr(), n, state_in, state_out, self.rms._moments_ptr,
                float(self.gamma), float(self.clip), float(self.epsilon), int(self.legacy_index_reset))
        return out if out.shape[0] == n else out[:n]


class RescaleNormalizer(BaseNormalizer):
    """Apply constant scaling (normalization.py:193-214)."""

    def __init__(self, coef=1.0):
        super().__init__()
        self.coef = coef

    def __call__(self, x):
        if not isinstance(x, torch.Tensor):
            x = np.asarray(x)
        return self.coef * x


class ImageNormalizer(RescaleNormalizer):
    """Scale image pixels from [0, 255] to [0, 1] (normalization.py:217-223)."""

    def __init__(self):
        super().__init__(1.0 / 255)


class ActionUnnormalizer(BaseNormalizer):
    """Assumes the policy's action is in [-1, 1]; maps it to the env's bounds (normalization.py:226-248)."""

    def __init__(self, action_space):
        super().__init__()
        assert hasattr(action_space, 'low') and hasattr(action_space, 'high'), 'action space must be a Box'
        low, high = np.asarray(action_space.low), np.asarray(action_space.high)
        self.mean = (low + high) / 2.0
        self.std = (high - low) / 2.0

    def __call__(self, action):
        if isinstance(action, torch.Tensor):
            f = dict(dtype=action.dtype, device=action.device)
            return torch.as_tensor(self.mean, **f) + action * torch.as_tensor(self.std, **f)
        return self.mean + np.asarray(action) * self.std
