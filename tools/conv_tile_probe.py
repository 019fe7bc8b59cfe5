import sys, torch
sys.path.insert(0, "/root/repo")
from pairnet_amd import hip
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for (H, W, Cin, Cout, K) in ((200, 334, 64, 64, 3), (100, 100, 64, 64, 7), (200, 334, 256, 256, 3)):
    x = torch.randn(1, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, K * K * Cin, generator=g) * 0.05).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    out = torch.empty(1, H, W, Cout, device=dev)
    for tile in (None, "128x64", "128"):
        run = lambda: hip.conv2d_nhwc(x, w, b, out, 1, H, W, Cin, Cout, K, K, K // 2, True, tile=tile)
        for _ in range(3): run()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(30): run()
        e.record(); torch.cuda.synchronize()
        us = 1e3 * s.elapsed_time(e) / 30
        print("%dx%d %d->%d k%d tile %-7s %7.1f us %6.1f TF" % (H, W, Cin, Cout, K, tile, us, 2.0 * H * W * Cin * Cout * K * K / us * 1e-6), flush=True)
