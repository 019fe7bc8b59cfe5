"""Tuning aid: time the native ResNet-50 at 800x1333 and list its kernels."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import ResNet50Hip, hip
dev = "cuda:0"
B, H, W = 1, 800, 1333
img = torch.randn(B, 3, H, W, device=dev)
nb = ResNet50Hip().to(dev)
for _ in range(3): nb(img)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): nb(img)
torch.cuda.synchronize()
print("native backbone: %.3f ms" % (1e3 * (time.perf_counter() - t) / 20))
hip.TIMER = hip.KernelTimer()
for _ in range(5): nb(img)
agg = hip.TIMER.summary(); hip.TIMER = None
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
    print("%-40s %4d launches %8.3f ms/iter %7.1f TF" % (k, v["launches"] // 5, v["ms"] / 5, v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0))
# per-launch list for one pass
hip.TIMER = hip.KernelTimer()
nb(img)
torch.cuda.synchronize()
for name, flops, nbytes, s, e in hip.TIMER.records:
    ms = s.elapsed_time(e)
    print("%-36s %8.1f us %6.1f TF  %.2f GF" % (name, 1e3 * ms, flops / (ms * 1e-3) / 1e12, flops / 1e9))
hip.TIMER = None
