"""Throughput of PSGTr.stream (pipeline + ResultStreamer behind one generator) at 800x1333."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import build_detector, pairnet_r50
dev = torch.device("cuda:0")
det = build_detector(pairnet_r50())
det.bbox_head.init_weights(seed=0)
det.to(dev)
H, W = 800, 1333
metas = [dict(img_shape=(H, W, 3), scale_factor=[2.083] * 4)]
g = torch.Generator().manual_seed(1)
pool = [torch.randn(1, 3, H, W, generator=g).to(dev) for _ in range(8)]
for copy in (False, True):
    for n in (30, 200):
        t = time.perf_counter()
        k = 0
        for results in det.stream(((pool[i % 8], metas) for i in range(n)), rescale=True, copy=copy):
            k += len(results)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print("copy=%s: %d images in %.2f s = %.1f images/s%s" % (
            copy, k, dt, k / dt, " (includes plan set-up and graph capture)" if n == 30 else ""), flush=True)
