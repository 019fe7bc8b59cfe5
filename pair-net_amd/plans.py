"""Bounded per-shape plan caches.

Every module of the path keeps its device buffers (and captured hipGraphs) in a plan per
(batch, feature shapes, pipeline slot, ...) so that the hot loop allocates nothing.  An
evaluation pass with keep-ratio resizing (configs/mask2former/pairnet.py:310-331, batch 1)
meets hundreds of distinct shapes; a CrossHead2 plan is ~1.3 GB at 800x1333, so the caches
are LRU-bounded: the least recently used plan is dropped (its buffers are freed once the
pipeline no longer holds it) when a new shape arrives at a full cache.
"""
from collections import OrderedDict

DEFAULT_MAX_PLANS = 16   # e.g. 4 pipeline slots x 4 shapes; ~20 GB of the 288 GB at full size


class PlanCache:
    """dict-like (`in`, `[]`, `[] =`, len, iteration in LRU order, values, clear)."""

    def __init__(self, max_plans=DEFAULT_MAX_PLANS):
        self._d = OrderedDict()
        self.max_plans = max_plans
        self.evictions = 0

    def __contains__(self, key):
        return key in self._d

    def __getitem__(self, key):
        self._d.move_to_end(key)
        return self._d[key]

    def __setitem__(self, key, value):
        self._d[key] = value
        self._d.move_to_end(key)
        while len(self._d) > max(1, self.max_plans):
            self._d.popitem(last=False)
            self.evictions += 1

    def __len__(self):
        return len(self._d)

    def __iter__(self):
        return iter(self._d)

    def values(self):
        return self._d.values()

    def items(self):
        return self._d.items()

    def clear(self):
        self._d.clear()
