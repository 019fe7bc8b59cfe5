"""Latency of the synchronous, reference-shaped call PSGTr.simple_test (fresh numpy arrays)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import build_detector, pairnet_r50
dev = torch.device("cuda:0")
det = build_detector(pairnet_r50())
det.bbox_head.init_weights(seed=0)
det.to(dev)
det.bbox_head.use_graphs = det.backbone.use_graphs = True
H, W = 800, 1333
metas = [dict(img_shape=(H, W, 3), scale_factor=[2.083] * 4, ori_shape=(384, 640, 3))]
g = torch.Generator().manual_seed(1)
pool = [torch.randn(1, 3, H, W, generator=g).to(dev) for _ in range(4)]
for i in range(6):
    r = det.simple_test(pool[i % 4], metas, rescale=True)
torch.cuda.synchronize()
n = 20
t = time.perf_counter()
for i in range(n):
    r = det.simple_test(pool[i % 4], metas, rescale=True)
dt = (time.perf_counter() - t) / n
print("simple_test: %.2f ms per call (masks %s %s)" % (1e3 * dt, r[0].masks.shape, r[0].masks.dtype))
t = time.perf_counter()
for i in range(n):
    feat = det.extract_feat(pool[i % 4])
    res = det.bbox_head.simple_test(feat, metas, rescale=True)
    torch.cuda.synchronize()
print("device part alone: %.2f ms per call" % (1e3 * (time.perf_counter() - t) / n))
# (History: a packed fetch into a FRESH array per call measured 18-20 ms against 14.7 for
# `.cpu()` -- first-touch page faults of 49 MB from the unpack threads; with arrays recycled
# through a weakref-finalised pool the call takes 8.4 ms, 12.0 when the caller keeps every result.)
t = time.perf_counter()
keep = []
for i in range(n):
    keep.append(det.simple_test(pool[i % 4], metas, rescale=True))
print("simple_test, caller keeps every result: %.2f ms per call" % (1e3 * (time.perf_counter() - t) / n))
