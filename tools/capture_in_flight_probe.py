"""Round-5's second "someone else's activity changes my numbers" phenomenon (labnotes R5.9), as
a reproducer that runs from a clean checkout: does capturing hipGraphs WHILE the 4-stream
pipeline is in flight still give a wrong image?

    python tools/capture_in_flight_probe.py unsafe 60      # capture wherever a graph is missing
    python tools/capture_in_flight_probe.py quiet 20       # the product's policy (control)
    python tools/capture_in_flight_probe.py unsafe+sync 60 # in flight, device wait kept in _capture
    PAIRNET_LIB=tools/gpubin/libpairnet_packed.so python tools/capture_in_flight_probe.py unsafe 60 fp32
        # round 5's conditions: a library WITH packed fp32 (python tools/build_variant.py packed
        # "-Xclang -target-feature -Xclang +packed-fp32-ops") and exact-fp32 GEMMs everywhere

A trial = 36 images (two distinct ones, ResNet-50 -> head -> get_bboxes) through
PipelinedHead(depth=4, a_streams=2) with backbone, stage and get_bboxes graphs on; after the 8th
image every captured graph is dropped (the head's and the backbone's `grid_reserve` hint changes:
both bake it into their graphs), so the following submissions re-capture with three batches in
flight.  `unsafe` replaces `plans.quiet` by "always" and takes the device wait out of
`CrossHead2._capture` (round 5's first design).  Every image's labels / rel_dists / panoptic map /
top-k list is compared bit for bit with the eager single-stream result.  The stream / event calls
the package issues while a stream of the process is capturing are logged for the first trial
(an event recorded on a capturing stream is a graph node, not a dependency: the judge's
hypothesis for R5.9).  Product imports only.
"""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pairnet_amd import CrossHead2, PipelinedHead, ResNet50Hip, pairnet_head_cfg  # noqa: E402
import pairnet_amd.plans as plans  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "unsafe"
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 40
arith = sys.argv[3] if len(sys.argv) > 3 else None     # "fp32": rounds 1-5's GEMMs everywhere
DEV = torch.device("cuda:0")
torch.cuda.set_device(DEV)

cfg = pairnet_head_cfg()
cfg.pop("type")
head = CrossHead2(**cfg)
head.init_weights(seed=0)
head.to(DEV)
if arith:
    head.gemm_arithmetic = arith
net = ResNet50Hip().to(DEV)
H, W = 800, 1333
g = torch.Generator().manual_seed(7)
imgs = [torch.randn(1, 3, H, W, generator=g).to(DEV) for _ in range(2)]
metas = [dict(img_shape=(H, W, 3), scale_factor=[2.083] * 4)]


def record_of(res, pl):
    r = res[0]
    return [t.clone() for t in (r[1], r[7], r[4], pl.topk_idx)]


head.use_graphs = net.use_graphs = False
eager = []
for im in imgs:
    res = head.simple_test_bboxes(net(im, slot=7), metas)
    torch.cuda.synchronize()
    eager.append(record_of(res, head._last_plan))
head.use_graphs = net.use_graphs = True

# ---- the policy under test ----------------------------------------------------------------
CAPTURING = [False]
LOG = collections.Counter()
LOGGING = [True]
orig_capture = CrossHead2._capture


def capture_no_wait(fn):
    dev = torch.cuda.current_device()
    cs = CrossHead2._capture_streams.get(dev)
    if cs is None:
        cs = CrossHead2._capture_streams[dev] = torch.cuda.Stream(dev)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cs):
        gr.capture_begin(capture_error_mode="thread_local")
        CAPTURING[0] = True
        try:
            fn()
        finally:
            CAPTURING[0] = False
            gr.capture_end()
    CrossHead2.captures += 1
    return gr


def capture_logged(fn):
    def inner():
        CAPTURING[0] = True
        try:
            fn()
        finally:
            CAPTURING[0] = False
    return orig_capture(inner)


KEEP = []       # a dropped graph stays alive until the trial's device wait (it may be executing)


def keeping(cap):
    def w(fn):
        gr = cap(fn)
        KEEP.append(gr)
        return gr
    return w


if mode.startswith("unsafe"):
    plans.quiet = lambda cur: True
    CrossHead2._capture = staticmethod(keeping(capture_logged if mode == "unsafe+sync"
                                               else capture_no_wait))
else:
    CrossHead2._capture = staticmethod(keeping(capture_logged))


def logged(cls, name):
    orig = getattr(cls, name)

    def w(self, *a, **k):
        if CAPTURING[0] and LOGGING[0]:
            st = a[0] if (a and isinstance(a[0], torch.cuda.Stream)) else \
                (self if isinstance(self, torch.cuda.Stream) else torch.cuda.current_stream())
            with torch.cuda.stream(st):
                capt = torch.cuda.is_current_stream_capturing()
            LOG["%s.%s on a %s stream" % (cls.__name__, name, "CAPTURING" if capt else "live")] += 1
        return orig(self, *a, **k)
    setattr(cls, name, w)


for c, n in ((torch.cuda.Event, "record"), (torch.cuda.Event, "query"),
             (torch.cuda.Event, "synchronize"), (torch.cuda.Stream, "wait_event"),
             (torch.cuda.Stream, "wait_stream"), (torch.cuda.Stream, "synchronize")):
    logged(c, n)

pipe = PipelinedHead(head, depth=4, a_streams=2)
net.grid_reserve = pipe.grid_reserve
order = [0, 1, 1, 0, 0, 1, 0, 1, 1, 0, 1, 0] * 3
bad_trials, bad_images, capture_subs, bad_with_capture = 0, 0, 0, 0
details = []
for t in range(trials):
    got, had_capture = [], []

    def take(res):
        got.append(record_of(res, head._last_plan))
        pipe.consumed(res)      # (the clones read slot buffers on this stream)

    for n, i in enumerate(order):
        if n == 8:      # drop every graph: the next submissions capture with batches in flight
            r = 64 if pipe.grid_reserve != 64 else 60
            pipe.grid_reserve = head.grid_reserve = net.grid_reserve = r
        c0 = CrossHead2.captures
        sl = pipe.count % len(pipe.streams_a)
        pipe.streams_a[sl].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(pipe.streams_a[sl]):
            res = pipe.submit(net(imgs[i], slot=sl), metas)
            if res is not None:
                take(res)
        had_capture.append(CrossHead2.captures - c0)
    while pipe.queue:
        take(pipe._finish(pipe.queue.pop(0)))
    torch.cuda.synchronize()
    del KEEP[:-64]
    LOGGING[0] = False
    wrong = [n for n, (i, r) in enumerate(zip(order, got))
             if not all(bool(torch.equal(a, b)) for a, b in zip(eager[i], r))]
    capture_subs += sum(1 for c in had_capture if c)
    if wrong:
        bad_trials += 1
        bad_images += len(wrong)
        # a capture in submission n can touch the results of images n-3 .. n (in flight then)
        near = [n for n in wrong if any(had_capture[max(0, n - 1):n + 4])]
        bad_with_capture += len(near)
        if len(details) < 12:
            details.append(dict(trial=t, wrong_images=wrong,
                                captures_per_submission=had_capture))
out = dict(mode=mode, gemm_arithmetic=head.gemm_arithmetic, library=os.environ.get("PAIRNET_LIB", "product"),
           trials=trials, images_per_trial=len(order), trials_with_a_wrong_image=bad_trials,
           wrong_images=bad_images, wrong_images_within_3_submissions_of_a_capture=bad_with_capture,
           submissions_with_a_capture=capture_subs, graphs_captured=CrossHead2.captures,
           calls_while_capturing_first_trial=dict(LOG), details=details,
           device=torch.cuda.get_device_name(0))
print(json.dumps(out))
