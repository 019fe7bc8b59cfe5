"""Two streams, different operands, the head's split-eligible launch shapes (positional addend,
batched Winograd form) mixed: are concurrent launches ever wrong?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pairnet_amd import hip
hip.lib()
DEV = "cuda:0"
torch.manual_seed(0)
def mk(j):
    d = {}
    d["x"] = torch.randn(21950, 256, device=DEV); d["pos"] = torch.randn(21950, 256, device=DEV)
    d["wvoa"] = torch.randn(544, 256, device=DEV) / 16; d["bvoa"] = torch.randn(544, device=DEV)
    d["w1"] = torch.randn(1024, 256, device=DEV) / 16; d["b1"] = torch.randn(1024, device=DEV)
    d["V"] = torch.randn(36, 4175, 256, device=DEV); d["U"] = torch.randn(36, 256, 256, device=DEV) / 16
    d["mf"] = torch.randn(66800, 256, device=DEV); d["wm"] = torch.randn(256, 256, device=DEV) / 16
    return d
def run(d, out):
    hip.linear(d["x"], d["wvoa"], d["bvoa"], out["voa"], aadd=d["pos"], aadd_from_col=256)
    hip.linear(d["x"], d["w1"], d["b1"], out["h"], relu=True)
    hip.gemm(d["V"], d["U"], out["m"], M=4175, N=256, K=256, lda=256, ldw=256, ldc=256, batch=36,
             sA=4175 * 256, sW=256 * 256, sC=4175 * 256)
    hip.linear(d["mf"], d["wm"], None, out["o"])
def outs():
    return dict(voa=torch.empty(21950, 544, device=DEV), h=torch.empty(21950, 1024, device=DEV),
                m=torch.empty(36, 4175, 256, device=DEV), o=torch.empty(66800, 256, device=DEV))
data = [mk(0), mk(1)]
with hip.split_gemm(True):
    want = [outs(), outs()]
    for j in range(2):
        run(data[j], want[j])
    torch.cuda.synchronize()
    ss = [torch.cuda.Stream(), torch.cuda.Stream()]
    bad = {}
    for rep in range(40):
        got = [outs(), outs()]
        torch.cuda.synchronize()
        for j in range(2):
            with torch.cuda.stream(ss[j]):
                run(data[j], got[j])
        torch.cuda.synchronize()
        for j in range(2):
            for k in got[j]:
                if not torch.equal(got[j][k], want[j][k]):
                    dd = (got[j][k] - want[j][k]).abs()
                    nz = dd.nonzero()
                    bad[k] = bad.get(k, 0) + 1
                    if bad[k] <= 2:
                        print("rep", rep, "stream", j, k, "wrong elements", len(nz), "max %.3e" % dd.max().item(),
                              "index range", nz.min(0).values.tolist(), nz.max(0).values.tolist(), flush=True)
print("wrong outputs by launch:", bad)
