"""pn_msda_f32 alone at the 800x1333 encoder shape: init-grid offsets and init + N(0, 8 px)."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
if os.environ.get("LIB"):          # an alternative build of the library (tools/README.md)
    hip.LIB_PATH = os.path.abspath(os.environ["LIB"])
dev = "cuda:0"
shapes = [(25, 42), (50, 84), (100, 167)]
SN = sum(h * w for h, w in shapes)
g = torch.Generator().manual_seed(0)
B = int(os.environ.get("BATCH", 1))
voa = torch.randn(B, SN, 544, generator=g).to(dev)
th = torch.arange(8, dtype=torch.float32) * (2.0 * math.pi / 8)
grid = torch.stack([th.cos(), th.sin()], -1)
grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(8, 1, 1, 2).repeat(1, 3, 4, 1)
for i in range(4):
    grid[:, :, i, :] *= i + 1
voa[..., 256:448] = grid.reshape(-1).to(dev)
voa[..., 448:] = 0
out = torch.empty(B, SN, 256, device=dev)
ref = torch.empty(B, SN, 256, device=dev)
spread = voa.clone()
spread[..., 256:448] += 8.0 * torch.randn(B, SN, 192, generator=g).to(dev)
alg = 4.0 * B * SN * (256 + 8 * 3 * 4 * 3 + 256)
VARIANTS = (("one-shot 8/CU", 0), ("one-shot 5/CU", hip.MSDA_LOW_OCCUPANCY), ("persistent", hip.MSDA_PERSISTENT),
            ("persistent x2", hip.MSDA_PERSISTENT_BATCHED))
for name, v, flags in [("%-10s %s" % (dn, vn), dv, fl) for rep in range(2)
                       for dn, dv in (("init grid", voa), ("N(0,8px)", spread)) for vn, fl in VARIANTS]:
    run = lambda: hip.msda(v, 544, v.view(-1)[256:], 544, out, B, shapes, flags=flags)
    for _ in range(5):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(100):
        run()
    e.record(); torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / 100
    print("%-24s %7.2f us  %6.0f GB/s algorithmic  checksum %.6e" % (name, us, alg / us * 1e-3, float(out.double().sum())))
