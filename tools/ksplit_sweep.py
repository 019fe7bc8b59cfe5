"""Split-K factor sweep on the mid-size GEMM shapes of one image -> triplets step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev = "cuda:0"
SHAPES = [(4200, 256, 1024, 1, 6), (4200, 1024, 256, 1, 6), (1050, 256, 256, 16, 5), (16700, 512, 128, 1, 4),
          (1050, 2048, 512, 1, 3), (16700, 128, 512, 1, 3), (16700, 256, 512, 1, 2), (4200, 128, 128, 16, 3),
          (1050, 512, 2048, 1, 2), (273, 512, 512, 16, 2), (4200, 512, 1024, 1, 1), (1050, 256, 2048, 1, 1),
          (21950, 256, 256, 1, 6), (21950, 544, 256, 1, 6), (21950, 1024, 256, 1, 6), (21950, 256, 1024, 1, 6)]
g = torch.Generator().manual_seed(0)
scratch = torch.empty(64 * 1024 * 1024, device=dev)
reserve = int(os.environ.get("RESERVE", 64))
tot_auto = tot_best = 0.0
for M, N, K, b, cnt in SHAPES:
    A = torch.randn(b * M, K, generator=g).to(dev)
    W = torch.randn(b * N, K, generator=g).to(dev) * 0.05
    bias = torch.randn(N, generator=g).to(dev)
    C = torch.empty(b * M, N, device=dev)
    res = {}
    for S in (0, 1, 2, 3, 4, 6, 8):
        if S > max(1, K // 32):
            continue
        def run():
            with hip.reserve_slots(reserve):
                hip.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias, batch=b, sA=M * K,
                         sW=N * K, sC=M * N, scratch=scratch, ksplit=S, relu=True)
        for _ in range(3):
            run()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(30):
            run()
        e.record()
        torch.cuda.synchronize()
        res[S] = 1e3 * s.elapsed_time(e) / 30
    best = min((v, k) for k, v in res.items() if k)
    fl = 2.0 * M * N * K * b
    tot_auto += cnt * res[0]
    tot_best += cnt * best[0]
    print("%6dx%4dx%4d b%-2d x%d tiles %5d  auto %6.1f us (%5.1f TF) | " % (
        M, N, K, b, cnt, -(-M // 64) * -(-N // 64) * b, res[0], fl / res[0] * 1e-6) +
        " ".join("S%d %5.1f" % (k, v) for k, v in res.items() if k) + " | best S%d %5.1f TF" % (best[1], fl / best[0] * 1e-6),
        flush=True)
print("per image: auto %.1f us, best %.1f us" % (tot_auto, tot_best))
