"""Tuning aid: time the native Swin backbone at 800x1333 and list its kernels.
usage: swin_probe.py [T|B|L] [batch]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import SwinTransformerHip, swin_backbone_cfg, hip
dev = "cuda:0"
variant = sys.argv[1] if len(sys.argv) > 1 else "L"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = "f32"
H, W = 800, 1333
cfg = swin_backbone_cfg(variant)
cfg.pop("type")
img = torch.randn(B, 3, H, W, device=dev)
nb = SwinTransformerHip(**cfg).to(dev)
for _ in range(2): nb(img)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): nb(img)
torch.cuda.synchronize()
print("Swin-%s %s batch %d: %.3f ms / forward (eager)" % (variant, mode, B, 1e3 * (time.perf_counter() - t) / 5))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    nb(img)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        nb(img)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): g.replay()
torch.cuda.synchronize()
print("graph replay: %.3f ms" % (1e3 * (time.perf_counter() - t) / 5))
hip.TIMER = hip.KernelTimer()
for _ in range(2): nb(img)
agg = hip.TIMER.summary(); hip.TIMER = None
tot = sum(v["ms"] for v in agg.values()) / 2
print("sum of kernel times %.3f ms" % tot)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
    print("%-40s %4d launches %8.3f ms/iter %7.1f TF %7.0f GB/s" % (
        k, v["launches"] // 2, v["ms"] / 2, v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0,
        v.get("bytes", 0) / (v["ms"] * 1e-3) / 1e9 if v["ms"] else 0))
if "--list" in sys.argv:
    hip.TIMER = hip.KernelTimer()
    nb(img)
    torch.cuda.synchronize()
    for name, flops, nbytes, s, e in hip.TIMER.records:
        ms = s.elapsed_time(e)
        print("%-36s %8.1f us %6.1f TF  %.2f GF" % (name, 1e3 * ms, flops / (ms * 1e-3) / 1e12, flops / 1e9))
    hip.TIMER = None
