"""Tuning aid: deformable-attention sampling (pn_msda_f32) at the 800x1333 pyramid for mmcv's
init offsets and for spread (learned-like) offsets.  (Round 2 also timed an LDS-window
variant with this script -- tools/attic/msda_window_variant.hip.txt: 119 us against 50 us for
the L2-gather kernel at the init offsets, 190-200 us with spread offsets; not adopted.)"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev = "cuda:0"
shapes = [(25, 42), (50, 84), (100, 167)]
n = sum(h * w for h, w in shapes)
def T(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)*1e3/reps
th = torch.arange(8, dtype=torch.float32) * (2.0 * math.pi / 8)
grid = torch.stack([th.cos(), th.sin()], -1)
grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(8, 1, 1, 2).repeat(1, 3, 4, 1)
for i in range(4): grid[:, :, i, :] *= i + 1
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for spread in (0.0, 2.0, 8.0, 32.0):
    off = grid[None, None].repeat(B, n, 1, 1, 1, 1) + spread * torch.randn(B, n, 8, 3, 4, 2)
    voa = torch.cat([torch.randn(B, n, 256), off.reshape(B, n, -1), torch.zeros(B, n, 96)], -1).contiguous().to(dev)
    out = torch.empty(B, n, 256, device=dev)
    row = []
    us = T(lambda: hip.msda(voa, 544, voa.view(-1)[256:], 544, out, B, shapes))
    row.append("%6.1f us = %5.0f GB/s algorithmic" % (us, B * 70.24e6 / us / 1e3))
    print("offsets = init grid + N(0, %4.1f px): %s" % (spread, " | ".join(row)))
