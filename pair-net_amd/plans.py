"""Per-slot buffer arenas and bounded plan caches: shape-polymorphic execution.

The reference's test loop (tools/test.py:199-267) feeds one image per step through
`Resize(keep_ratio) -> Pad(size_divisor=1)` (configs/mask2former/pairnet.py:310-321): every
distinct original size is a distinct tensor shape, hundreds per PSG split.  The modules of
this path therefore do NOT own buffers per shape.  Each pipeline slot owns ONE flat device
buffer (`Arena`), sized for the largest shape met so far (its "envelope": the element-wise
maximum of the size-driving dimensions), and a plan for a (shape, slot) is nothing but a set
of VIEWS carved out of that buffer in a fixed order (`Carver`).  Plans of different shapes
alias each other -- a slot holds one image at a time -- so:

  * device memory depends on the largest shape, not on how many shapes were seen;
  * the first sight of a shape costs ~100 tensor views: no allocation, no host
    synchronisation, no eviction;
  * per-shape device state is only what is read-only and small enough to share between the
    slots (the sine position tables, kept in a small LRU of their own by the head).

An arena grows only when the envelope grows (a handful of times per process: landscape,
then portrait); growing waits for the device and drops the slot's plans (their views -- and
the hipGraphs captured on them -- point into the old buffer).

`PlanCache` keeps the plans (views + captured graphs) LRU-bounded; an evicted plan may
still be executing on some stream, so it is parked with an event per stream it ran on and
released once those have fired (`reap`): eviction never blocks the host either.
"""
from collections import OrderedDict

import torch

DEFAULT_MAX_PLANS = 64   # e.g. 4 pipeline slots x 16 shapes; plans are views (+ hipGraphs)
ALIGN = 256


# ---- when a hipGraph may be captured -----------------------------------------------------
# Measured in round 5 (LABNOTES R5.9): capturing a graph while OTHER streams of the process are
# executing (the pipeline in flight: a backbone graph replaying on one stream while the head's
# stage graph is captured for the next) corrupted ~10 % of the runs of a 12-image test -- one
# image's result slightly off, only with backbone AND head graphs on, never with a device wait
# in front of the capture (0 / 20), never eagerly (0 / 28).  Capture therefore happens only at
# QUIET points: nothing this package queued on any stream other than the caller's current one is
# unfinished; the capture then waits for the device (which at a quiet point costs at most
# the caller's own queued work).  In flight, a (shape, slot) without a graph simply runs eagerly
# -- within 0.5 % of the replay rate -- until a quiet point comes (warm-up, a synchronous caller,
# `PSGTr.warm_graphs`).
_LAST = {}       # (device index, stream handle) -> event behind this package's last launches there


def note_use(stream):
    """Call after queueing work on `stream`: (re-)records that stream's event.  (An event of our
    own rather than `stream.query()`: the legacy default stream's query is not a reliable idle
    test on this stack -- it was seen False right after a device wait.)"""
    key = (stream.device_index, stream.cuda_stream)
    ev = _LAST.get(key)
    if ev is None:
        ev = _LAST[key] = torch.cuda.Event()
    ev.record(stream)


def quiet(current):
    """True when the work this package queued on every stream of `current`'s device other than
    `current` has finished."""
    for (dev, handle), ev in list(_LAST.items()):     # (a snapshot: another thread may add a stream)
        if dev == current.device_index and handle != current.cuda_stream and not ev.query():
            return False
    return True


class ArenaOverflow(RuntimeError):
    pass


class Carver:
    """Bump allocator over an arena's buffer: `E(*shape)` -> float32 view, `E.i64 / .i32 /
    .u8 / .f64 / .bool(*shape)` likewise.  On a `None` buffer (measuring mode) it hands out
    meta tensors and only counts bytes; `used` is the running total either way."""

    def __init__(self, buf, capacity):
        self.buf, self.capacity, self.used = buf, capacity, 0

    def _take(self, shape, dtype):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        off = self.used
        self.used = (off + nbytes + ALIGN - 1) // ALIGN * ALIGN
        if self.buf is None:
            return torch.empty(tuple(int(d) for d in shape), dtype=dtype, device="meta")
        if self.used > self.capacity:
            raise ArenaOverflow()
        return self.buf[off:off + nbytes].view(dtype).view(tuple(int(d) for d in shape))

    def __call__(self, *shape):
        return self._take(shape, torch.float32)

    def f64(self, *shape):
        return self._take(shape, torch.float64)

    def i64(self, *shape):
        return self._take(shape, torch.int64)

    def i32(self, *shape):
        return self._take(shape, torch.int32)

    def u8(self, *shape):
        return self._take(shape, torch.uint8)


class Arena:
    """One flat device buffer; `carve(layout, dims, measure)` returns what `layout(E)` builds
    on views of it.  `dims`: the tuple of integers that drive the layout's sizes (batch,
    heights, widths ...: every buffer size must be non-decreasing in each of them);
    `measure(dims)` -> bytes the layout needs for `dims` (a `Carver(None, 0)` pass).  When the
    layout does not fit, the envelope becomes max(envelope, dims) element-wise and the buffer
    is re-allocated for it: `generation` changes and `on_grow()` tells the owner to drop the
    views it handed out."""

    def __init__(self, device, on_grow=None, slack=1.0):
        self.device, self.on_grow, self.slack = torch.device(device), on_grow, slack
        self.buf, self.capacity, self.envelope, self.generation = None, 0, None, 0
        self.grows = 0

    def reserve(self, dims, measure):
        """Make room for every shape whose dims are <= `dims` element-wise."""
        dims = tuple(int(d) for d in dims)
        env = dims if self.envelope is None or len(self.envelope) != len(dims) else \
            tuple(max(a, b) for a, b in zip(self.envelope, dims))
        need = max(measure(env), measure(dims))
        if env == self.envelope and need <= self.capacity:
            return
        self.envelope = env
        if need <= self.capacity:
            return
        # kernels queued on any stream (and captured graphs) may still use the old buffer
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        self.buf = None
        if self.on_grow is not None:
            self.on_grow(self)
        self.capacity = int(need * self.slack + ALIGN - 1) // ALIGN * ALIGN
        self.buf = torch.empty(self.capacity, dtype=torch.uint8, device=self.device)
        self.generation += 1
        self.grows += 1

    def carve(self, layout, dims, measure):
        for _ in range(2):
            if self.buf is not None:
                try:
                    return layout(Carver(self.buf, self.capacity))
                except ArenaOverflow:
                    pass
            self.reserve(dims, measure)
        raise RuntimeError("arena: the layout does not fit the size it measured")


def measure_bytes(layout):
    """Bytes `layout(E)` carves (run on meta tensors)."""
    c = Carver(None, 0)
    layout(c)
    return c.used


class PlanCache:
    """dict-like LRU (`in`, `[]`, `[] =`, len, iteration in LRU order, values, clear, `drop`).
    Evicted values that expose `busy_events()` (events recorded behind their last use on every
    stream they ran on) are parked until those events have fired, so that a captured hipGraph
    or a buffer is never released under a kernel that still uses it -- without a host wait."""

    def __init__(self, max_plans=DEFAULT_MAX_PLANS):
        self._d = OrderedDict()
        self.max_plans = max_plans
        self.evictions = 0
        self._parked = []

    def __contains__(self, key):
        return key in self._d

    def __getitem__(self, key):
        self._d.move_to_end(key)
        return self._d[key]

    def get(self, key, default=None):
        return self[key] if key in self._d else default

    def __setitem__(self, key, value):
        self._d[key] = value
        self._d.move_to_end(key)
        while len(self._d) > max(1, self.max_plans):
            _, old = self._d.popitem(last=False)
            self.evictions += 1
            self._park(old)

    def _park(self, value):
        busy = getattr(value, "busy_events", None)
        events = list(busy()) if busy is not None else []
        if events:
            self._parked.append((value, events))
        self.reap()

    def reap(self):
        """Release parked values whose streams have passed their last use."""
        self._parked = [(v, ev) for v, ev in self._parked if not all(e.query() for e in ev)]
        return len(self._parked)

    def drop(self, pred):
        """Remove (and park) every entry whose key satisfies `pred`."""
        for k in [k for k in self._d if pred(k)]:
            self._park(self._d.pop(k))

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
