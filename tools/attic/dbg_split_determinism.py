"""Is k_gemm_split deterministic (alone, back to back, on two streams at once)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pairnet_amd import hip
hip.lib()
DEV = "cuda:0"
torch.manual_seed(0)
for (M, N, K) in ((21950, 1024, 256), (21950, 544, 256), (66800, 256, 256), (16700, 512, 1024), (4175, 256, 256)):
    x, w, b = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / 16, torch.randn(N, device=DEV)
    outs = []
    with hip.split_gemm(True):
        for i in range(6):
            o = torch.empty(M, N, device=DEV)
            hip.linear(x, w, b, o, relu=True)
            outs.append(o)
        torch.cuda.synchronize()
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        o1 = [torch.empty(M, N, device=DEV) for _ in range(4)]
        o2 = [torch.empty(M, N, device=DEV) for _ in range(4)]
        torch.cuda.synchronize()
        for i in range(4):
            with torch.cuda.stream(s1):
                hip.linear(x, w, b, o1[i], relu=True)
            with torch.cuda.stream(s2):
                hip.linear(x, w, b, o2[i], relu=True)
        torch.cuda.synchronize()
        same2 = all(torch.equal(outs[0], o) for o in o1 + o2)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    print(M, N, K, "back to back identical:", same, " two streams identical:", same2,
          " max err vs fp64 %.3e" % (outs[0].double() - ref).abs().max().item(), flush=True)
