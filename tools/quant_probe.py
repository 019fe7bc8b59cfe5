"""Tuning aid: tile-count quantisation of the persistent GEMM (TFLOP/s vs M)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev = "cuda:0"
def T(fn, n=30):
    for _ in range(5): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) * 1e3 / n
for N, K in ((1024, 256), (256, 1024), (256, 256)):
    w = torch.randn(N, K, device=dev) * 0.1
    for M in (16384, 20480, 21950, 24576, 32768, 65536):
        x = torch.randn(M, K, device=dev); o = torch.empty(M, N, device=dev)
        us = T(lambda: hip.linear(x, w, None, o))
        tiles = ((M + 63) // 64) * (N // 64)
        print("N=%4d K=%4d M=%6d tiles=%6d (%.2f per WG) %7.1f us %6.1f TF" % (N, K, M, tiles, tiles / 1024, us, 2.0 * M * N * K / us / 1e6))
