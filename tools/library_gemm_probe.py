"""The hand-written fp32-MFMA GEMM against the library (torch.nn.functional.linear = rocBLAS /
hipBLASLt fp32, TF32 off) on the path's shapes: back-to-back launches, HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from pairnet_amd import hip

torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda:0"
SHAPES = [("encoder FFN-1", 21950, 1024, 256), ("encoder FFN-2", 21950, 256, 1024),
          ("[value|offsets|logits]", 21950, 544, 256), ("encoder out-proj", 21950, 256, 256),
          ("R50 stage-3 conv3", 4200, 1024, 256), ("R50 stage-3 conv1", 4200, 256, 1024),
          ("R50 stage-2 conv3", 16700, 512, 128), ("R50 stage-4 conv1", 1050, 512, 2048),
          ("mask feature", 66800, 256, 256), ("C5 input conv", 1050, 256, 2048),
          ("C4 input conv", 4200, 256, 1024)]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


scratch = torch.empty(9 * 1024 * 1024, device=dev)
for name, M, N, K in SHAPES:
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    t_lib = timeit(lambda: F.linear(x, w, b))
    t_own = timeit(lambda: hip.linear(x, w, b, out, scratch=scratch))
    err = float((out - F.linear(x, w, b)).abs().max())
    gf = 2.0 * M * N * K * 1e-9
    print("%-24s M=%-6d N=%-5d K=%-5d  library %6.1f us %6.1f TF | this repo %6.1f us %6.1f TF | max diff %.1e"
          % (name, M, N, K, t_lib, gf / t_lib * 1e3, t_own, gf / t_own * 1e3, err), flush=True)
