"""pn_linear_res_ln_f32 (one launch) against pn_gemm_f32 + pn_layernorm_f32 on the encoder's two
N = 256 shapes, eager, back to back (tools/gemm_ln_probe.py; run on the GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
if os.environ.get("LIB"):          # an alternative build of the library (tools/README.md)
    hip.LIB_PATH = os.path.abspath(os.environ["LIB"])
dev = "cuda:0"
torch.manual_seed(0)
M = int(os.environ.get("ROWS", 21950))
for K in (256, 1024):
    x = torch.randn(M, K, device=dev)
    if K == 1024:
        x = torch.relu(x)          # the FFN hidden map is a ReLU output
    w = torch.randn(256, K, device=dev) * 0.05
    b, res = torch.randn(256, device=dev), torch.randn(M, 256, device=dev)
    g, be = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev)
    pre, y1, y2 = (torch.empty(M, 256, device=dev) for _ in range(3))

    def pair():
        hip.linear(x, w, b, pre, res=res)
        hip.layernorm(pre, g, be, y1)

    def gemm_only():
        hip.linear(x, w, b, pre, res=res)

    def fused():
        hip.linear_res_ln(x, w, b, res, g, be, y2)
    for name, fn in (("gemm + layernorm", pair), ("gemm alone", gemm_only), ("fused", fused),
                     ("gemm + layernorm", pair), ("fused", fused)):
        for _ in range(5):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(60):
            fn()
        e.record(); torch.cuda.synchronize()
        us = 1e3 * s.elapsed_time(e) / 60
        print("M %d K %4d  %-18s %7.2f us  %6.1f TFLOP/s" % (M, K, name, us, 2.0 * M * 256 * K / us * 1e-6))
    print("  bitwise equal:", bool(torch.equal(y1, y2)))
