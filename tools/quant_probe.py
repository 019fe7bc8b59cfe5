"""Tile-count quantisation of the mid-size GEMMs: time each shape at its real M and at the
nearest M whose 64x64 tile count is a whole number of rounds of the persistent grid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev = "cuda:0"
SHAPES = [(4200, 1024, 256), (4200, 256, 1024), (16700, 512, 128), (16700, 128, 512),
          (1050, 2048, 512), (1050, 512, 2048), (16700, 256, 512), (4200, 512, 1024),
          (21950, 256, 256), (21950, 1024, 256)]
g = torch.Generator().manual_seed(0)
scratch = torch.empty(64 * 1024 * 1024, device=dev)
reserve = int(os.environ.get("RESERVE", 0))


def timeit(M, N, K):
    A = torch.randn(M, K, generator=g).to(dev)
    W = torch.randn(N, K, generator=g).to(dev) * 0.05
    bias = torch.randn(N, generator=g).to(dev)
    C = torch.empty(M, N, device=dev)

    def run():
        with hip.reserve_slots(reserve):
            hip.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias, scratch=scratch, relu=True)
    for _ in range(3):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(40):
        run()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / 40


for M, N, K in SHAPES:
    nt = -(-N // 64)
    rows = -(-M // 64)
    line = "%6dx%4dx%4d tiles %5d:" % (M, N, K, rows * nt)
    for r in (rows, rows - 1, rows - 2, rows + 1, rows - rows % 16, rows - rows % 16 + 16):
        if r <= 0:
            continue
        m = r * 64 if r != rows else M
        us = timeit(m, N, K)
        line += "  rows %3d (%5d t) %5.1f us %5.1f TF |" % (r, r * nt, us, 2.0 * m * N * K / us * 1e-6)
    print(line, flush=True)
