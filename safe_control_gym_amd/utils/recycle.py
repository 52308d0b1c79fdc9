"""Fresh outputs without an allocation per call: output sets recycled once nobody refers to them.

The reference returns new arrays from ``env.step()`` and from its normalisers (quadrotor.py:558, normalization.py:125-133), and its
RL loops rely on it (controllers/ppo/ppo.py:259-276 keeps ``obs`` across the next ``step()``).  On the device a new tensor per
output costs ~2 us of host time and every view of it 1-1.5 us more -- at the reference's batch sizes, where the host is the bound,
that is half the cost of a step.  A ``RecyclingPool`` keeps whole output sets (one allocation, its views made once) and hands a set
out again only when it is at rest: the use count of its storage (every tensor and view of it, whoever holds it) and the Python
reference counts of the objects that are handed out (a caller holding the very object adds no storage reference) are what they were
when the set was made.  That is the caching allocator's rule -- a block is reused once its tensor died -- applied to the tensors
themselves; anything anybody can still read is never written again.  Errors are one-sided: an unexpected reference (a debugger, a
reference cycle waiting for the garbage collector) keeps a set busy, it never frees one.
"""
import sys

import torch

_use_count = torch._C._storage_Use_Count
_refcount = sys.getrefcount


class RecyclingPool:
    """``factory() -> (payload, watched, storage)``: ``payload`` is what ``take()`` returns (any object holding the set's tensors and
    views; holding it keeps the set busy too), ``watched`` a tuple of the Python objects that get handed out or bound somewhere (all
    of them reachable from ``payload``), ``storage`` the ``UntypedStorage`` every tensor of the set lives in.  ``cap`` = the most sets the pool may hold."""

    __slots__ = ('factory', 'cap', 'slots', 'pos', 'wait', 'backoff')

    def __init__(self, factory, cap):
        self.factory, self.cap = factory, int(cap)
        self.slots, self.pos, self.wait, self.backoff = [], 0, 0, 0

    @staticmethod
    def _counts(slot):
        return (_use_count(slot[2]), _refcount(slot[0]), *map(_refcount, slot[1]))      # storage, the payload object, each watched object

    def _grow(self):
        payload, watched, storage = self.factory()
        slot = [payload, watched, storage._cdata, None, storage]
        del payload, watched, storage
        slot[3] = self._counts(slot)                   # the set at rest
        self.slots.append(slot)
        return slot[0]

    def take(self):
        """-> the payload of a set nobody refers to (the least recently used first), of a new set while the pool may grow, or None:
        the caller keeps whatever it is handed (e.g. appends every reward to a list) -- allocate as before then; the pool looks
        again after a growing number of calls."""
        if self.wait:
            self.wait -= 1
            return None
        slots, counts = self.slots, self._counts
        k, p = len(slots), self.pos
        for _ in range(k):
            slot = slots[p]
            p = p + 1 if p + 1 < k else 0
            if counts(slot) == slot[3]:
                self.pos, self.backoff = p, 0
                return slot[0]
        if k < self.cap:
            self.pos = 0
            return self._grow()
        self.backoff = min(4096, max(32, 2 * self.backoff))
        self.wait = self.backoff
        return None

    def __len__(self):
        return len(self.slots)
