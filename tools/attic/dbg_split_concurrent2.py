"""Stream 1: split GEMMs; stream 2: a default-path kernel on other data.  Who is disturbed?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pairnet_amd import hip
hip.lib()
DEV = "cuda:0"
torch.manual_seed(0)
M, N, K = 21950, 1024, 256
x1, w1 = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / 16
x2, w2 = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / 16
w3 = torch.randn(256, 1024, device=DEV) / 32
g, b = torch.randn(256, device=DEV), torch.randn(256, device=DEV)
res = torch.randn(M, 256, device=DEV)
h2 = torch.randn(M, 1024, device=DEV)

def split_job(o):
    hip.gemm(x1, w1, o, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, split=True)

def tile_job(o):
    hip.gemm(x2, w2, o, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, split=False)

def rowln_job(o):
    hip.linear_res_ln(h2, w3, None, res, g, b, o[:, :256]) if False else hip.linear_res_ln(h2, w3, torch.zeros(256, device=DEV), res, g, b, o[:, :256].contiguous() if False else o256)

o256 = torch.empty(M, 256, device=DEV)
for name, job in (("k_gemm_tile", tile_job),):
    want1, want2 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    split_job(want1); job(want2)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    o1 = [torch.empty(M, N, device=DEV) for _ in range(10)]
    o2 = [torch.empty(M, N, device=DEV) for _ in range(10)]
    for i in range(10):
        with torch.cuda.stream(s1):
            split_job(o1[i])
        with torch.cuda.stream(s2):
            job(o2[i])
    torch.cuda.synchronize()
    b1 = sum(not torch.equal(o, want1) for o in o1)
    b2 = sum(not torch.equal(o, want2) for o in o2)
    print("split beside", name, ": wrong split outputs", b1, "/10, wrong", name, "outputs", b2, "/10", flush=True)
    for o in o2:
        if not torch.equal(o, want2):
            d = (o - want2).abs(); nz = d.nonzero()
            print("   ", name, "wrong elements", len(nz), "max", d.max().item(), "rows", nz[:, 0].min().item(), nz[:, 0].max().item(),
                  "cols", nz[:, 1].min().item(), nz[:, 1].max().item())
            break
    for o in o1:
        if not torch.equal(o, want1):
            d = (o - want1).abs(); nz = d.nonzero()
            print("    split wrong elements", len(nz), "max", d.max().item(), "rows", nz[:, 0].min().item(), nz[:, 0].max().item(),
                  "cols", nz[:, 1].min().item(), nz[:, 1].max().item())
            break
