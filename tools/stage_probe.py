"""Tuning aid: stage A / stage B alone (graph replay and eager), no overlap."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import CrossHead2, pairnet_head_cfg
dev = torch.device("cuda:0")
cfg = pairnet_head_cfg(); cfg.pop("type")
head = CrossHead2(**cfg); head.init_weights(seed=0); head.to(dev)
shapes = [(200, 334), (100, 167), (50, 84), (25, 42)]
NB = int(os.environ.get("BATCH", "1"))
feats = [torch.relu(torch.randn(NB, c, h, w)).to(dev) for c, (h, w) in zip((256, 512, 1024, 2048), shapes)]
metas = [dict(img_shape=(800, 1333, 3), scale_factor=[2.083] * 4)] * NB
def T(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t) / n
head.use_graphs = True
for _ in range(3): head.forward(feats, metas)
pl = head._last_plan
print("graph replay: stage A %.3f ms, stage B %.3f ms" % (T(lambda: pl.graph_a.replay()), T(lambda: pl.graph_b.replay())))
head.use_graphs = False
print("eager:        stage A %.3f ms, stage B %.3f ms" % (T(lambda: head._stage_a(feats, pl)), T(lambda: head._stage_b(pl))))
outs = head._outputs(pl)
print("get_bboxes eager %.3f ms" % T(lambda: head.get_bboxes(*outs, metas)))
if "--kernels" in sys.argv:
    from pairnet_amd import hip
    for name, fn in (("A", lambda: head._stage_a(feats, pl)), ("B", lambda: head._stage_b(pl))):
        hip.TIMER = hip.KernelTimer()
        for _ in range(3): fn()
        agg = hip.TIMER.summary(); hip.TIMER = None
        tot = sum(v["ms"] for v in agg.values()) / 3
        print("stage %s: %d launches, %.3f ms of kernel time (events around each launch)" % (
            name, sum(v["launches"] for v in agg.values()) // 3, tot))
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            print("  %-44s %4d x %8.1f us = %7.3f ms  %6.1f TF %6.0f GB/s" % (
                k, v["launches"] // 3, 1e3 * v["ms"] / v["launches"], v["ms"] / 3,
                v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0,
                v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] else 0))
