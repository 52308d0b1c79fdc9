"""utils/recycle.py on host tensors (the counters are those of any torch storage): a set is handed out again exactly when nobody
refers to any part of it -- the object itself, a view derived from it, a container that holds it -- and never otherwise."""
import gc

import torch

from safe_control_gym_amd.utils.recycle import RecyclingPool


def _factory():
    arena = torch.zeros(64 + 16, dtype=torch.uint8)
    planes = arena[:64].view(torch.float32).view(4, 4)
    flags = arena[64:].view(torch.bool)
    obs = planes.T
    return (planes, flags, obs), (planes, flags, obs), arena.untyped_storage()


def _ptr(payload):
    return payload[0].data_ptr()


def test_a_set_at_rest_is_reused_and_a_held_one_is_not():
    pool = RecyclingPool(_factory, cap=4)
    a = pool.take()
    pa = _ptr(a)
    del a
    b = pool.take()
    assert _ptr(b) == pa and len(pool) == 1                    # nobody held it: the same set again
    c = pool.take()
    assert _ptr(c) != pa and len(pool) == 2                    # `b` is held: another set
    obs = b[2]
    del b
    d = pool.take()
    assert _ptr(d) not in (pa,) or len(pool) == 3              # the very object handed out is still held
    assert _ptr(d) != pa
    view = obs[:2]                                             # only a view DERIVED from it survives
    del obs
    e = pool.take()
    assert _ptr(e) != pa
    del view
    f = pool.take()
    assert _ptr(f) == pa                                       # ... and once that is gone the set comes back
    del c, d, e, f


def test_a_container_or_a_cycle_keeps_its_set_busy_until_it_is_gone():
    pool = RecyclingPool(_factory, cap=2)
    a = pool.take()
    pa = _ptr(a)
    box = {'flags': a[1]}
    box['self'] = box                                          # a reference cycle: only the garbage collector frees it
    del a
    b = pool.take()
    assert _ptr(b) != pa
    pb = _ptr(b)
    del b, box
    gc.collect()
    got = {_ptr(pool.take()), _ptr(pool.take())}               # (each payload dropped at once: both sets are at rest)
    assert got <= {pa, pb}


def test_an_exhausted_pool_returns_none_and_backs_off():
    pool = RecyclingPool(_factory, cap=2)
    held = [pool.take(), pool.take()]
    assert pool.take() is None and pool.wait == 32             # the caller allocates for itself for the next 32 calls
    for _ in range(32):
        assert pool.take() is None
    assert pool.wait == 0
    assert pool.take() is None and pool.wait == 64             # still hoarded: twice as long
    held.clear()
    for _ in range(64):
        assert pool.take() is None                             # (not even looked at while backing off)
    assert pool.take() is not None and pool.backoff == 0       # free again


def test_written_values_survive_while_referenced():
    pool = RecyclingPool(_factory, cap=3)
    kept = []
    for k in range(50):
        s = pool.take()
        if s is None:
            s = _factory()[0]
        s[0].fill_(float(k))
        if k % 7 == 0:
            kept.append((k, s[2]))                             # the transposed view, as env.step() hands out its obs
        del s
    for k, obs in kept:
        assert bool((obs == float(k)).all())
