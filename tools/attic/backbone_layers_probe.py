"""Tuning aid: per-launch time / rate of the native ResNet-50 at 800x1333 (HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import ResNet50Hip, hip
dev = "cuda:0"
nb = ResNet50Hip().to(dev)
img = torch.randn(1, 3, 800, 1333, device=dev)
for _ in range(3): nb(img)
torch.cuda.synchronize()
N = 10
hip.TIMER = t = hip.KernelTimer()
for _ in range(N): nb(img)
torch.cuda.synchronize()
per = len(t.records) // N
tot = 0.0
for i in range(per):
    ms = sum(t.records[j * per + i][3].elapsed_time(t.records[j * per + i][4]) for j in range(N)) / N
    name, fl, by = t.records[i][0], t.records[i][1], t.records[i][2]
    tot += ms
    print("%2d %-36s %7.1f us %6.1f TF %7.1f GB/s  (%.2f GF, %.1f MB)" % (i, name, ms * 1e3, fl / ms / 1e9, by / ms / 1e6, fl / 1e9, by / 1e6))
print("sum of timed launches %.3f ms" % tot)
