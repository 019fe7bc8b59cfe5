import sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from pairnet_amd import hip
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for B, H, W in ((1, 800, 1333), (2, 800, 1333), (1, 37, 53)):
    img = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    ref = F.relu(F.conv2d(img.double(), w.double(), b.double(), stride=2, padding=3))
    wp = torch.zeros(64, 160); wp[:, :147] = w.reshape(64, 147)
    Ho, Wo = ref.shape[-2:]
    out = torch.empty(B, Ho, Wo, 64, device=dev)
    run = lambda: hip.stem7x7s2(img_d, wp_d, b_d, out, B, H, W)
    img_d, wp_d, b_d = img.to(dev), wp.to(dev), b.to(dev)
    for _ in range(3): run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(30): run()
    e.record(); torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / 30
    err = float((out.permute(0, 3, 1, 2).cpu().double() - ref).abs().max() / ref.abs().max())
    print("B%d %dx%d: %.1f us  %.1f TFLOP/s  rel err %.2e" % (B, H, W, us, 2.0 * B * Ho * Wo * 64 * 147 / us * 1e-6, err), flush=True)
