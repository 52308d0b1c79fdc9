"""Sharding the env batch across GPUs (one process per GPU) and gathering episode returns.

Environments never interact, so rank r simply owns the contiguous block
``[r * N/W, (r+1) * N/W)`` of global env indices -- the block split ``SubprocVecEnv`` makes with
``np.array_split`` (env_wrappers/vectorized_env/subproc_vec_env.py:28,52).  The step path has no
collective.  The one exchange is the analogue of gathering ``info['episode']['r']`` into the
return queue (record_episode_statistics.py:155-158): an all-gather of the per-env episode returns
(``torch.distributed``; backend ``nccl`` is RCCL over xGMI on ROCm, ``gloo`` on CPU for tests).
"""
import os

import torch
import torch.distributed as dist


def shard_bounds(total_envs, world_size, rank):
    """[lo, hi) of the global env indices owned by ``rank`` (np.array_split semantics: the first
    ``total % world`` ranks get one extra env)."""
    base, extra = divmod(int(total_envs), int(world_size))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, world, local_rank)."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('SCG_SINGLE_DEVICE') == '1':
        # rehearsal on a one-GPU box: every rank drives device 0 and the collectives run over gloo
        local_rank, backend = 0, (backend or 'gloo')
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend)
    return rank, world, local_rank


def make_sharded(make_fn, total_envs, rank=0, world_size=1, **kwargs):
    """Build this rank's shard: ``make_fn(num_envs=hi-lo, env_index_base=lo, **kwargs)``."""
    lo, hi = shard_bounds(total_envs, world_size, rank)
    return make_fn(num_envs=hi - lo, env_index_base=lo, **kwargs)


def all_gather_episode_returns(local_returns, total_envs=None, force_collective=False):
    """[n_local] -> [total] on every rank (rank order = global env order).

    Uneven shards are padded to the largest one so that a single all_gather_into_tensor suffices.  A world of one rank
    returns a copy without touching the backend, unless ``force_collective`` (tests: the RCCL call on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        return local_returns.clone()
    world, rank = dist.get_world_size(), dist.get_rank()
    n_local = local_returns.shape[0]
    if total_envs is None:
        sizes = torch.tensor([n_local], device=local_returns.device, dtype=torch.int64)
        all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(all_sizes, sizes)
        counts = [int(s.item()) for s in all_sizes]
    else:
        counts = [shard_bounds(total_envs, world, r)[1] - shard_bounds(total_envs, world, r)[0] for r in range(world)]
    width = max(counts)
    dev = local_returns.device
    # gloo (CPU tests, or several ranks sharing one GPU) gathers host tensors; nccl = RCCL gathers in place on device
    via_host = dist.get_backend() == 'gloo' and dev.type == 'cuda'
    send = local_returns.cpu() if via_host else local_returns
    if n_local != width:
        padded = torch.zeros(width, dtype=send.dtype, device=send.device)
        padded[:n_local] = send
        send = padded
    out = torch.empty(world * width, dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(out, send.contiguous())
    if via_host:
        out = out.to(dev)
    if all(c == width for c in counts):
        return out
    return torch.cat([out[r * width:r * width + counts[r]] for r in range(world)])


class PendingGather:
    """An all-gather of episode returns in flight (``begin_all_gather_episode_returns``); ``wait()`` -> the [total] tensor."""

    def __init__(self, out, work, deferred=None, keep=None):
        self._out, self._work, self._deferred, self._keep = out, work, deferred, keep     # keep: the snapshot being sent

    def wait(self):
        if self._deferred is not None:                      # no asynchronous path for this backend / shard shape
            self._out, self._deferred = self._deferred(), None
        elif self._work is not None:
            self._work.wait()                               # RCCL: the current stream waits, the host does not
            self._work = None
        return self._out


def begin_all_gather_episode_returns(local_returns, total_envs=None, force_collective=False):
    """Start the all-gather on a SNAPSHOT of the returns and hand back a ``PendingGather``: the step kernels issued next keep
    updating the plane while RCCL moves the snapshot over xGMI on its own stream (the exchange costs the step path nothing).
    Even shards over RCCL run asynchronously; gloo, uneven shards and a world of one complete inside ``wait()``."""
    snap = local_returns.clone()
    ready = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_collective)
    if ready and dist.get_backend() == 'nccl' and total_envs is not None and total_envs % dist.get_world_size() == 0:
        out = torch.empty(dist.get_world_size() * snap.shape[0], dtype=snap.dtype, device=snap.device)
        return PendingGather(out, dist.all_gather_into_tensor(out, snap, async_op=True), keep=snap)
    return PendingGather(None, None, deferred=lambda: all_gather_episode_returns(snap, total_envs=total_envs,
                                                                                 force_collective=force_collective))


def all_reduce_stats(stats):
    """Sum a small stats tensor (e.g. the kernel's [8] batch sums) over ranks."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    return stats
