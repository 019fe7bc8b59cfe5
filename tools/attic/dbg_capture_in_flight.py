import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import golden
import test_production_gpu as T
from pairnet_amd import PipelinedHead
DEV="cuda:0"
fx = golden("e2e_image_full")
det = T._detector(fx, "r50")
img, metas = T._image(fx)
imgs = [img[i:i + 1].contiguous().to(DEV) for i in range(2)]
head, net = det.bbox_head, det.backbone
eager = []
for im in imgs:
    r = head.simple_test_bboxes(net(im), metas[:1])[0]
    eager.append([t.clone() for t in (r[1], r[7], r[4], head._last_plan.topk_idx)])
head.use_graphs = net.use_graphs = True
mode = sys.argv[1] if len(sys.argv) > 1 else ""
if mode == "noA": head.STAGE_A_GRAPHS = 0
if mode == "nobb": net.use_graphs = False
if mode in ("cssync", "cswait", "devsync_after"):
    import pairnet_amd.head as HH
    orig = HH.CrossHead2._capture
    def cap2(fn):
        g = orig(fn)
        cs = HH.CrossHead2._capture_streams[torch.cuda.current_device()]
        if mode == "cssync":
            cs.synchronize()
        elif mode == "cswait":
            torch.cuda.current_stream().wait_stream(cs)
        else:
            torch.cuda.synchronize()
        return g
    HH.CrossHead2._capture = staticmethod(cap2)
if mode == "headonly":
    net.use_graphs = False
if mode == "bbonly":
    head.use_graphs = False
if mode == "eager":
    head.use_graphs = net.use_graphs = False
if mode == "defer":
    head.defer_first_replay = True
    net.defer_first_replay = True
if mode == "cursync":
    import pairnet_amd.head as HH
    orig0 = HH.CrossHead2._capture
    def cap3(fn):
        torch.cuda.current_stream().synchronize()
        return orig0(fn)
    HH.CrossHead2._capture = staticmethod(cap3)
if mode == "sync":
    import pairnet_amd.head as HH
    orig = HH.CrossHead2._capture
    def cap(fn):
        torch.cuda.synchronize()
        return orig(fn)
    HH.CrossHead2._capture = staticmethod(cap)
NSUB = [0]
if mode.startswith("stall"):
    # no graphs at all; the HOST stalls where a capture would have stalled it (in front of the
    # backbone, stage A, stage B and get_bboxes launches of submissions 4..7): does a late host
    # alone perturb a result?
    import time
    import pairnet_amd.head as HH
    import pairnet_amd.backbone as BB
    head.use_graphs = net.use_graphs = False
    ms = float(mode[5:] or 3) * 1e-3
    def stalled(fn):
        def w(*a, **k):
            if 4 <= NSUB[0] < 8:
                t = time.perf_counter()
                while time.perf_counter() - t < ms:
                    pass
            return fn(*a, **k)
        return w
    HH.CrossHead2._stage_a = stalled(HH.CrossHead2._stage_a)
    HH.CrossHead2._stage_b = stalled(HH.CrossHead2._stage_b)
    HH.CrossHead2._get_bboxes_all = stalled(HH.CrossHead2._get_bboxes_all)
    BB.ResNet50Hip._run = stalled(BB.ResNet50Hip._run)
pipe = PipelinedHead(head, depth=4, a_streams=2)
order = [0, 1, 1, 0, 0, 1, 0, 1, 1, 0, 1, 0] * 3
got = []
def take(res):
    pl = head._last_plan
    got.append([t.clone() for t in (res[0][1], res[0][7], res[0][4], pl.topk_idx)] + [pl.slot, [(e["calls"], e["graph"] is not None) for e in pl.graphs_a.values()]])
for i in order:
    NSUB[0] = pipe.count
    sl = pipe.count % len(pipe.streams_a)
    pipe.streams_a[sl].wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(pipe.streams_a[sl]):
        res = pipe.submit(net(imgs[i], slot=sl), metas[:1])
        if res is not None:
            take(res)
while pipe.queue:
    take(pipe._finish(pipe.queue.pop(0)))
torch.cuda.synchronize()
for n, (i, g) in enumerate(zip(order, got)):
    if all(bool(torch.equal(a, b)) for a, b in zip(eager[i], g[:4])): continue
    print(n, "img", i, "slot", g[4], "entries", g[5], "same", [bool(torch.equal(a, b)) for a, b in zip(eager[i], g[:4])])
