"""Tuning aid: the R50 backbone's 1x1 convolutions at 800x1333 (row-major GEMMs with the
residual + ReLU epilogue where the layer has one) on 64x64 against 128x64 tiles, sustained
(200 back-to-back launches, ReLU-like inputs).  Run on the GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev = "cuda:0"


def T(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


shapes = [(66800, 64, 256, False), (66800, 256, 64, True), (66800, 128, 256, False),
          (16700, 512, 128, True), (16700, 128, 512, False),
          (4200, 1024, 256, True), (4200, 256, 1024, False),
          (1050, 2048, 512, True), (1050, 512, 2048, False)]
for M, N, K, has_res in shapes:
    x = torch.relu(torch.randn(M, K, device=dev))
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    res = torch.relu(torch.randn(M, N, device=dev)) if has_res else None
    o = torch.empty(M, N, device=dev)
    sc = torch.empty(16 * 1024 * 1024, device=dev)
    row = []
    for name, kw in (("auto", {}), ("tile64", dict(force="tile64")), ("t128x64", dict(force="tile128x64")),
                     ("auto", {})):
        us = T(lambda: hip.linear(x, w, b, o, res=res, relu=not has_res, relu_after=has_res,
                                  scratch=sc, **kw))
        row.append("%s %6.1fus %5.1fTF" % (name, us, 2.0 * M * N * K / us / 1e6))
    nbytes = 4.0 * (M * K + N * K + M * N * (2 if has_res else 1))
    print("%6d %5d %5d res=%d | %s | %.0f MB" % (M, N, K, has_res, " | ".join(row), nbytes / 1e6))
