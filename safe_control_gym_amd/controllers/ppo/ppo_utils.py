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

_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)


class PPOBuffer:
    """(T, N, ...) float32 device tensors under the reference's keys (ppo_utils.py:308-336).

    ``env=<batched env>`` (new) lays ``obs``, ``rew`` and the ``done`` flags out as the ENGINE'S OWN planes -- observation slabs
    ``[T + 1][obs_dim][plane_stride]``, reward / done planes ``[T][plane_stride]`` -- so that the step kernel writes a step's
    outputs straight into row t (``env.step(act, out=buf.step_out())``): ``push`` then only carries the keys the env does not
    produce (``act``, ``v``, ``logp``, ``terminal_v`` ...).  ``buf.obs`` / ``buf.rew`` / ``buf.mask`` keep their ``(T, N, ...)`` shapes: ``obs``
    and ``rew`` are zero-copy views of the planes (a transposed view for ``obs``, as ``env.state`` is), ``mask`` = ``1 - done`` is
    formed on demand (``compute_returns_and_advantages(..., dones=buf.done)`` reads the byte planes directly).  Slab t of the
    observation planes is the observation the policy saw at step t; slab t + 1 is where step t's kernel writes the next one; at
    the end of a rollout ``start(obs)`` copies the last observation into slab 0 of the next (one slab per T steps)."""

    def __init__(self, obs_space, act_space, max_length, batch_size, device='cuda', env=None):
        self.max_length, self.batch_size = max_length, batch_size
        self.device = torch.device(device) if env is None else env.device
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
        self.planes = env is not None
        self._lib = None
        if self.planes:
            if env.num_envs != N or len(obs_dim) != 1 or obs_dim[0] != env.obs_dim:
                raise ValueError('[ERROR] PPOBuffer(env=...): batch_size / obs_space do not match the env.')
            self._ld, self._od = env.plane_stride, env.obs_dim
        self.reset()

    _ENGINE_KEYS = ('obs', 'rew', 'mask')

    def reset(self):
        self._push_rows, self._push_plans = {}, {}
        for k, info in self.scheme.items():
            if self.planes and (k in self._ENGINE_KEYS or k == 'act'):
                continue
            init = info.get('init', torch.zeros)
            if self.planes and info['vshape'][2:] == (1,):
                # every per-step scalar as (T, plane_stride) planes whose (T, N, 1) view is what callers see: the returns / GAE
                # kernel then reads the whole buffer in place (one row stride for all of its inputs)
                self.__dict__[k] = init((self.max_length, self._ld), dtype=torch.float32, device=self.device)[:, :self.batch_size].unsqueeze(-1)
                continue
            self.__dict__['_mask' if k == 'mask' else k] = init(info['vshape'], dtype=torch.float32, device=self.device)
        if self.planes:
            T, N, ld = self.max_length, self.batch_size, self._ld
            self.obs_planes = torch.zeros((T + 1, self._od, ld), dtype=torch.float32, device=self.device)
            self.rew_planes = torch.zeros((T, ld), dtype=torch.float32, device=self.device)
            self.done_planes = torch.zeros((T, ld), dtype=torch.bool, device=self.device)
            self.__dict__['obs'] = self.obs_planes[:T, :, :N].permute(0, 2, 1)                 # (T, N, obs_dim), zero copy
            self.__dict__['rew'] = self.rew_planes[:, :N].unsqueeze(-1)                        # (T, N, 1), zero copy
            # the actions as planes too: a policy that writes act_planes[t] (coalesced) hands the env its action planes in place
            # (env.step(buf.act_planes[t])); push({'act': rows}) transposes into them
            self.act_planes = torch.zeros((T, self.scheme['act']['vshape'][2], ld), dtype=torch.float32, device=self.device)
            self.__dict__['act'] = self.act_planes[:, :, :N].permute(0, 2, 1)                  # (T, N, act_dim), zero copy
            # the per-step rows as ready-made views: indexing a tensor costs 2-3 us of host time, a rollout step needs eight of them,
            # and below ~2^17 envs the host is what bounds an eager rollout
            self._obs_rows, self._rew_rows = self.obs_planes.unbind(0), self.rew_planes.unbind(0)
            self._done_rows, self._act_rows = self.done_planes.unbind(0), self.act_planes.unbind(0)
            self._obs_views = [row[:, :N].T for row in self._obs_rows]
            self._step_outs = {}
        self._dev = self.__dict__['v'].device                   # with its index, as a policy's tensors carry it
        self.t = 0

    # ---- the engine-written rows (env=...) ---------------------------------------------------------------------------
    @property
    def done(self):
        """(T, N) bool: the step kernel's done flags (``mask = 1 - done``)."""
        return self.done_planes[:, :self.batch_size]

    @property
    def mask(self):
        if self.planes:
            return (~self.done).to(torch.float32).unsqueeze(-1)
        return self.__dict__['_mask']

    @mask.setter
    def mask(self, value):
        self.__dict__['_mask'] = value

    def start(self, obs):
        """Begin a rollout: ``obs`` [N, obs_dim] (the env's current observation, e.g. the previous rollout's last one) becomes
        slab 0.  Returns the [N, obs_dim] view of slab 0."""
        self.t = 0
        self.obs_planes[0, :, :self.batch_size].copy_(torch.as_tensor(obs, dtype=torch.float32, device=self.device).T)
        return self.obs_at(0)

    def obs_at(self, t):
        """[N, obs_dim] view of observation slab t (t = max_length: the observation after the last step, for the bootstrap value)."""
        return self._obs_views[t]

    def obs_row(self, t):
        """Observation slab t as planes [obs_dim, plane_stride] (``obs_normalizer(raw, out=buf.obs_row(t + 1))``)."""
        return self._obs_rows[t]

    def rew_row(self, t):
        """Reward row t [plane_stride] (``reward_normalizer(rew, done, out=buf.rew_row(t))``)."""
        return self._rew_rows[t]

    def done_row(self, t):
        return self._done_rows[t]

    def act_row(self, t):
        """Action planes of step t [act_dim, plane_stride]: a policy that writes them hands the env its action in place."""
        return self._act_rows[t]

    def step_out(self, t=None, obs=True, reward=True):
        """The ``out=`` argument of ``env.step`` for step t (default: the current row): the next observation into slab t + 1, reward
        and done into row t.  ``obs=False`` / ``reward=False`` leave that output with the env (a normaliser between the env and the
        buffer then writes the row itself: ``obs_normalizer(next_obs, out=buf.obs_planes[t + 1])``,
        ``reward_normalizer(rew, done, out=buf.rew_planes[t])``)."""
        t = self.t if t is None else t
        key = (t, obs, reward)
        out = self._step_outs.get(key)
        if out is None:                                      # (built once per row: env.step only reads it)
            out = {'done': self._done_rows[t]}
            if obs:
                out['obs'] = self._obs_rows[t + 1]
            if reward:
                out['reward'] = self._rew_rows[t]
            self._step_outs[key] = out
        return out

    def _row(self, k, t):
        """Row t of key k's (T, N, ...) tensor as a ready-made view (indexing a tensor costs 2-3 us of host time, a push has six or
        seven keys); remembered per tensor OBJECT: ``buf.ret = ...`` (controllers/ppo/ppo_utils.py:443-444 assigns) makes new ones."""
        assert k in self.keys
        if self.planes and k in self._ENGINE_KEYS:
            raise KeyError('[ERROR] PPOBuffer(env=...): %r is written by the step kernel (env.step(act, out=buf.step_out())), '
                           'not pushed.' % k)
        name = '_mask' if k == 'mask' else k
        data = self.__dict__[name]
        rows = self._push_rows.get(name)
        if rows is None or rows[0] is not data:
            rows = self._push_rows[name] = (data, data.unbind(0))
        return name, data, rows[1][t]

    def _push_plan(self, t, keys):
        """Row t's destinations for this key order as the dst half of a ``scg_copy_segment`` array (the src half is filled per push):
        (keys, [(name, tensor object)], segment array or None, its elements, [(N, d)])."""
        owners, segs, elems, shapes = [], None, [], []
        rows = []
        for k in keys:
            name, data, row = self._row(k, t)
            owners.append((name, data))
            rows.append(row)
        if 1 <= len(keys) <= _capi.SCG_MAX_COPY_SEGMENTS and all(r.ndim == 2 and min(r.stride()) >= 0 for r in rows):
            segs = (_capi.ScgCopySegment * len(keys))()
            for j, row in enumerate(rows):
                g = segs[j]
                g.dst, g.rows, g.cols = row.data_ptr(), row.shape[0], row.shape[1]
                g.dst_row_stride, g.dst_col_stride = row.stride()
                elems.append(g)
                shapes.append((row.shape[0], row.shape[1]))
        return keys, owners, segs, elems, shapes

    def push(self, batch):
        """One step of data: values are [N, ...] tensors (or arrays); no host round trip for tensors.  With ``env=...`` the
        engine-written keys (``obs``, ``rew``, ``mask``) are refused here: they arrive through ``step_out()``.

        The per-step case -- every value a float32 tensor on the buffer's device, [N, d] (any strides: a policy's output, the
        [N, obs_dim] view of SoA planes that ``env.step()`` returns) or [N] for a (T, N, 1) key -- is ONE launch
        (``scg_copy_segments``: up to 8 strided copies whose descriptors travel as kernel arguments) instead of one ``copy_`` per
        key; anything else (host arrays, other dtypes or shapes) takes the per-key path."""
        t = self.t
        keys = tuple(batch)
        plan = self._push_plans.get(t)
        if plan is None or plan[0] != keys or any(self.__dict__[name] is not data for name, data in plan[1]):
            plan = self._push_plans[t] = self._push_plan(t, keys)
        dev = self._dev
        if plan[2] is not None and _RAW_STREAM is not None and dev.type == 'cuda' and torch.cuda.current_device() == dev.index:
            f32, elems, shapes = torch.float32, plan[3], plan[4]
            j = 0
            for v in batch.values():
                if type(v) is not torch.Tensor or v.dtype != f32 or v.device != dev:
                    break
                n, d = shapes[j]
                sh = v.shape
                if len(sh) == 2 and sh[0] == n and sh[1] == d:
                    rs, cs = v.stride()
                elif len(sh) == 1 and d == 1 and sh[0] == n:
                    rs, cs = v.stride(0), 1
                else:
                    break
                if rs < 0 or cs < 0:
                    break
                g = elems[j]
                g.src, g.src_row_stride, g.src_col_stride = v.data_ptr(), rs, cs
                j += 1
            else:
                if self._lib is None:
                    self._lib = _capi.load()
                status = self._lib.scg_copy_segments(plan[2], j, _RAW_STREAM(dev.index))
                if status != 0:
                    _capi.check(status, 'scg_copy_segments')
                self.t = (t + 1) % self.max_length
                return
        for k, v in batch.items():
            row = self._row(k, t)[2]
            if type(v) is torch.Tensor and v.dtype == torch.float32 and v.device == dev:
                row.copy_(v if v.shape == row.shape else v.reshape(row.shape))
            else:
                row.copy_(torch.as_tensor(v, dtype=torch.float32, device=self.device).reshape(row.shape))
        self.t = (t + 1) % self.max_length

    def _flat(self, k):
        info = self.scheme[k]
        x = self.mask if k == 'mask' else self.__dict__[k]
        return x.reshape(-1, *info['vshape'][2:])              # a copy for the transposed obs view of the plane layout

    def get(self, device=None):
        out = {}
        for k in self.scheme:
            data = self._flat(k)
            out[k] = data if device is None else data.to(device)
        return out

    def sample(self, indices):
        """Rows of the flattened (T * N, ...) data: gathered with (t, env) index pairs, so the transposed observation view of the
        plane layout is never materialised."""
        idx = torch.as_tensor(indices, device=self.device)
        t, i = torch.div(idx, self.batch_size, rounding_mode='floor'), idx % self.batch_size
        return {k: (self.mask if k == 'mask' else self.__dict__[k])[t, i] for k in self.scheme}

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


def _rows(x, T, N, name, dtypes=(torch.float32,)):
    """(T, N, 1) / (T, N) device tensor -> its (T, N) view (no copy when it is T rows of N consecutive elements, whatever the row
    stride: row-slices of wider (T, ld) planes are read in place)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        raise _capi.ScgError('%s must be a ROCm device tensor: the returns/GAE kernel has no CPU path' % name)
    x = x.reshape(T, N)
    if x.dtype not in dtypes:
        x = x.to(dtypes[0])
    if N > 1 and x.stride(1) != 1:
        x = x.contiguous()
    return x


def compute_returns_and_advantages(rews, vals, masks, terminal_vals=0, last_val=0, gamma=0.99, use_gae=False,
                                   gae_lambda=0.95, normalize_advantages=False, eps=1e-6, dones=None):
    """Same contract as ppo_utils.py:428-456 on (T, N, 1) device tensors; returns (rets, advs) of that shape.

    Like the reference, ``rews`` is updated in place with ``gamma * terminal_vals`` (:443).  With
    ``normalize_advantages=True`` the advantages are additionally standardised on the device,
    ``(adv - mean) / (std + eps)`` as ppo.py:288 does on the host.

    ``dones`` (new; then ``masks`` may be None): the step kernel's (T, N) bool / uint8 done flags (``PPOBuffer(env=...).done``) in
    place of the float ``masks = 1 - done``: the kernel reads the bytes and forms the mask itself.  Inputs that are row-slices of
    wider planes ((T, ld) storage: the plane-layout ``PPOBuffer``) are read in place when they all share the row stride; the
    results then have that layout too.
    """
    lib = _capi.load()
    T, N = rews.shape[:2]
    rew = _rows(rews, T, N, 'rews')
    val = _rows(vals, T, N, 'vals')
    done = mask = None
    if dones is not None:
        done = _rows(dones, T, N, 'dones', (torch.bool, torch.uint8))
    else:
        mask = _rows(masks, T, N, 'masks')
    dev = rew.device
    tv = None
    if isinstance(terminal_vals, torch.Tensor):
        tv = _rows(terminal_vals, T, N, 'terminal_vals')
    elif np.any(np.asarray(terminal_vals) != 0):
        tv = torch.as_tensor(np.broadcast_to(np.asarray(terminal_vals, dtype=np.float32).reshape(-1), (T * N,)).copy(),
                             device=dev).reshape(T, N)
    # one row stride for every (T, N) input (the C ABI has one ld); inputs that disagree are compacted
    ins = [t for t in (rew, val, mask, done, tv) if t is not None]
    ld = ins[0].stride(0) if T > 1 else N
    if T > 1 and any(t.stride(0) != ld for t in ins):
        ld = N
        rew, val = rew.contiguous(), val.contiguous()
        mask = None if mask is None else mask.contiguous()
        done = None if done is None else done.contiguous()
        tv = None if tv is None else tv.contiguous()
    if isinstance(last_val, torch.Tensor):
        lv = last_val.to(device=dev, dtype=torch.float32).reshape(N).contiguous()
    else:
        lv = torch.as_tensor(np.broadcast_to(np.asarray(last_val, dtype=np.float32).reshape(-1), (N,)).copy(), device=dev)
    rets = torch.empty((T, ld), dtype=torch.float32, device=dev)[:, :N]
    advs = torch.empty((T, ld), dtype=torch.float32, device=dev)[:, :N]
    stats = torch.zeros(2, dtype=torch.float64, device=dev) if normalize_advantages else None
    stream = _capi.current_stream(dev)
    p = _capi.ptr
    with torch.cuda.device(dev):
        rew_out = rew if tv is not None else None                    # in-place, as the reference does
        _capi.check(lib.scg_gae(p(rew), p(val), p(mask), p(done), p(tv), p(lv), p(rets), p(advs), p(rew_out), p(stats), T, N, ld,
                                float(gamma), float(gae_lambda), int(bool(use_gae)), stream), 'scg_gae')
        if normalize_advantages:
            _capi.check(lib.scg_normalize_advantages(p(advs), p(stats), T, N, ld, float(eps), stream),
                        'scg_normalize_advantages')
    if tv is not None and rew.data_ptr() != rews.data_ptr():
        rews.copy_(rew.reshape(rews.shape))
    return rets.unsqueeze(-1), advs.unsqueeze(-1)
