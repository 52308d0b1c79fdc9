"""Device-resident rollout storage and returns / GAE for the PPO-style callers of the batched env.

Mirror of the two pieces of the reference's ``controllers/ppo/ppo_utils.py`` that sit directly on the
``env.step()`` output: ``PPOBuffer`` (:282-407, same scheme keys, ``push`` /
``get`` / ``sample`` / ``sampler``) and ``compute_returns_and_advantages`` (:428-456).  The reference copies every
step to host NumPy arrays of shape (T, N, ...) and walks T in Python; here the (T, N) planes stay on the GPU
(``env.step`` already returns device tensors) and the backward recursion is one HIP kernel launch
(``scg_gae`` in ``include/scg_rollout.h``), one lane per env.
"""

import numpy as np
import torch

from safe_control_gym_amd import _capi


class PPOBuffer:
    """(T, N, ...) float32 device tensors under the reference's keys (ppo_utils.py:308-336)."""

    def __init__(self, obs_space, act_space, max_length, batch_size, device='cuda'):
        self.max_length, self.batch_size = max_length, batch_size
        self.device = torch.device(device)
        T, N = max_length, batch_size
        obs_dim = tuple(obs_space.shape)
        act_dim = act_space.shape[0] if hasattr(act_space, 'shape') and len(act_space.shape) else act_space.n
        self.scheme = {
            'obs': {'vshape': (T, N, *obs_dim)}, 'act': {'vshape': (T, N, act_dim)},
            'rew': {'vshape': (T, N, 1)}, 'mask': {'vshape': (T, N, 1), 'init': torch.ones},
            'v': {'vshape': (T, N, 1)}, 'logp': {'vshape': (T, N, 1)}, 'ret': {'vshape': (T, N, 1)},
            'adv': {'vshape': (T, N, 1)}, 'terminal_v': {'vshape': (T, N, 1)},
        }
        self.keys = list(self.scheme.keys())
        self.reset()

    def reset(self):
        for k, info in self.scheme.items():
            init = info.get('init', torch.zeros)
            self.__dict__[k] = init(info['vshape'], dtype=torch.float32, device=self.device)
        self.t = 0

    def push(self, batch):
        """One step of data: values are [N, ...] tensors (or arrays); no host round trip for tensors."""
        for k, v in batch.items():
            assert k in self.keys
            shape = self.scheme[k]['vshape'][1:]
            self.__dict__[k][self.t].copy_(torch.as_tensor(v, dtype=torch.float32, device=self.device).reshape(shape))
        self.t = (self.t + 1) % self.max_length

    def get(self, device=None):
        out = {}
        for k, info in self.scheme.items():
            data = self.__dict__[k].reshape(-1, *info['vshape'][2:])
            out[k] = data if device is None else data.to(device)
        return out

    def sample(self, indices):
        idx = torch.as_tensor(indices, device=self.device)
        return {k: self.__dict__[k].reshape(-1, *info['vshape'][2:])[idx] for k, info in self.scheme.items()}

    def sampler(self, mini_batch_size, device=None, drop_last=True):
        total = self.max_length * self.batch_size
        perm = torch.randperm(total, device=self.device)
        full = total // mini_batch_size * mini_batch_size
        for b in perm[:full].reshape(-1, mini_batch_size):
            batch = self.sample(b)
            yield batch if device is None else {k: v.to(device) for k, v in batch.items()}
        if not drop_last and total % mini_batch_size:
            batch = self.sample(perm[full:])
            yield batch if device is None else {k: v.to(device) for k, v in batch.items()}


def random_sample(indices, batch_size, drop_last=True):
    """ppo_utils.py:410-426: a random permutation of ``indices`` cut into batches (the remainder kept unless ``drop_last``)."""
    indices = torch.as_tensor(indices)
    perm = indices[torch.randperm(indices.shape[0], device=indices.device)]
    full = perm.shape[0] // batch_size * batch_size
    for batch in perm[:full].reshape(-1, batch_size):
        yield batch
    if not drop_last and perm.shape[0] % batch_size:
        yield perm[full:]


def _plane(x, T, N, name):
    """(T, N, 1) / (T, N) float32 contiguous device tensor -> itself viewed as T planes of N."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        raise _capi.ScgError('%s must be a ROCm device tensor: the returns/GAE kernel has no CPU path' % name)
    x = x.reshape(T, N)
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.to(torch.float32).contiguous()
    return x


def compute_returns_and_advantages(rews, vals, masks, terminal_vals=0, last_val=0, gamma=0.99, use_gae=False,
                                   gae_lambda=0.95, normalize_advantages=False, eps=1e-6):
    """Same contract as ppo_utils.py:428-456 on (T, N, 1) device tensors; returns (rets, advs) of that shape.

    Like the reference, ``rews`` is updated in place with ``gamma * terminal_vals`` (:443).  With
    ``normalize_advantages=True`` the advantages are additionally standardised on the device,
    ``(adv - mean) / (std + eps)`` as ppo.py:288 does on the host.
    """
    lib = _capi.load()
    T, N = rews.shape[:2]
    rew = _plane(rews, T, N, 'rews')
    val = _plane(vals, T, N, 'vals')
    mask = _plane(masks, T, N, 'masks')
    dev = rew.device
    tv = None
    if isinstance(terminal_vals, torch.Tensor):
        tv = _plane(terminal_vals, T, N, 'terminal_vals')
    elif np.any(np.asarray(terminal_vals) != 0):
        tv = torch.as_tensor(np.broadcast_to(np.asarray(terminal_vals, dtype=np.float32).reshape(-1), (T * N,)).copy(),
                             device=dev).reshape(T, N)
    if isinstance(last_val, torch.Tensor):
        lv = last_val.to(device=dev, dtype=torch.float32).reshape(N).contiguous()
    else:
        lv = torch.as_tensor(np.broadcast_to(np.asarray(last_val, dtype=np.float32).reshape(-1), (N,)).copy(), device=dev)
    rets = torch.empty((T, N), dtype=torch.float32, device=dev)
    advs = torch.empty((T, N), dtype=torch.float32, device=dev)
    stats = torch.zeros(2, dtype=torch.float64, device=dev) if normalize_advantages else None
    stream = _capi.current_stream(dev)
    p = _capi.ptr
    with torch.cuda.device(dev):
        rew_out = rew if tv is not None else None                    # in-place, as the reference does
        _capi.check(lib.scg_gae(p(rew), p(val), p(mask), p(tv), p(lv), p(rets), p(advs), p(rew_out), p(stats), T, N, N,
                                float(gamma), float(gae_lambda), int(bool(use_gae)), stream), 'scg_gae')
        if normalize_advantages:
            _capi.check(lib.scg_normalize_advantages(p(advs), p(stats), T, N, N, float(eps), stream),
                        'scg_normalize_advantages')
    if tv is not None and rew.data_ptr() != rews.data_ptr():
        rews.copy_(rew.reshape(rews.shape))
    return rets.reshape(T, N, 1), advs.reshape(T, N, 1)
