"""Attention launches of one image (pn_attention_f32), back to back per shape."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to(dev)
tot = 0.0
# masked: False, or the fraction of keys a query may NOT attend to (random weights: ~0.5;
# a trained checkpoint attends inside the predicted mask, ~10 % of the pixels, in blobs)
for B, Q, Nk, masked, per_img in ((1, 100, 16700, 0.5, 3), (1, 100, 4200, 0.5, 3), (1, 100, 1050, 0.5, 3),
                                  (1, 100, 100, False, 15), (1, 100, 200, False, 6), (2, 100, 16700, 0.5, 0),
                                  (1, 300, 300, False, 0), (1, 200, 16700, 0.5, 0),
                                  (1, 100, 16700, 0.9, 0), (1, 100, 4200, 0.9, 0)):
    q, k, v = R(B * Q, 256), R(B * Nk, 256), R(B * Nk, 256)
    bits = rowall = None
    if masked:
        logits = R(B * Q, Nk)
        if masked > 0.5:   # contiguous foreground blob per query (10 % of the keys)
            n = int(Nk * (1 - masked))
            start = torch.randint(0, Nk - n, (B * Q, 1), generator=g).to(dev)
            ar = torch.arange(Nk, device=dev)[None]
            logits = torch.where((ar >= start) & (ar < start + n), 1.0, -1.0)
        bits = torch.empty(B * Q * ((Nk + 31) // 32), device=dev, dtype=torch.int32)
        rowall = torch.empty(B * Q, device=dev, dtype=torch.int32)
        hip.mask_pack(logits, bits, rowall, B * Q, Nk)
    scr = torch.empty(hip.attn_scratch_floats(B, Q, Nk), device=dev)
    out = torch.empty(B * Q, 256, device=dev)
    run = lambda: hip.attention(q, 256, k, 256, v, 256, bits, rowall, out, 256, scr, B, Q, Nk, 1 / math.sqrt(32))
    for _ in range(5):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    n = 100
    for _ in range(n):
        run()
    e.record()
    torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / n
    fl = 4.0 * B * 8 * Q * Nk * 32
    tot += per_img * us
    print("B%d Q%d Nk%-6d masked=%s: %7.2f us  %6.1f TFLOP/s" % (B, Q, Nk, masked, us, fl / us * 1e-6), flush=True)
print("per image (30 launches): %.1f us, avg %.2f us/launch" % (tot, tot / 30))
