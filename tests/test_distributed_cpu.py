"""world_size-2 gloo tests of the N>1 path (CPU): shard bounds + the all-gather of episode returns."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from safe_control_gym_amd import parallel
    r, w, _ = parallel.init_from_env(backend='gloo')
    lo, hi = parallel.shard_bounds(total, w, r)
    local = torch.arange(lo, hi, dtype=torch.float32) * 0.5 + 1.0          # "episode return" of global env i
    gathered = parallel.all_gather_episode_returns(local, total_envs=total)
    gathered2 = parallel.all_gather_episode_returns(local)                  # sizes discovered by a collective
    stats = parallel.all_reduce_stats(torch.tensor([float(hi - lo), float(local.sum())], dtype=torch.float64))
    q.put((rank, gathered.numpy(), gathered2.numpy(), stats.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(total, world=2):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_all_gather_even_shards():
    total = 64
    expect = np.arange(total, dtype=np.float32) * 0.5 + 1.0
    for rank, g, g2, stats in _run(total):
        assert np.array_equal(g, expect) and np.array_equal(g2, expect)
        assert stats[0] == total and stats[1] == expect.sum()


def test_all_gather_uneven_shards():
    total = 37
    expect = np.arange(total, dtype=np.float32) * 0.5 + 1.0
    for rank, g, g2, stats in _run(total):
        assert np.array_equal(g, expect) and np.array_equal(g2, expect)


def test_single_process_passthrough():
    from safe_control_gym_amd import parallel
    x = torch.arange(5, dtype=torch.float32)
    assert torch.equal(parallel.all_gather_episode_returns(x), x)
