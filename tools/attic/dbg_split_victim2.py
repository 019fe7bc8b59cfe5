"""Split GEMM loop on one stream; on another, ONE default-path kernel of the head's stage A at a
time on fixed inputs: which kernel's output changes?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pairnet_amd import hip
hip.lib()
DEV = "cuda:0"
torch.manual_seed(0)
SN, shapes = 21950, [(25, 42), (50, 84), (100, 167)]
x = torch.randn(SN, 256, device=DEV); pos = torch.randn(SN, 256, device=DEV)
wvoa = torch.randn(544, 256, device=DEV) / 16; bvoa = torch.randn(544, device=DEV) * 0.1
w1 = torch.randn(1024, 256, device=DEV) / 16; b1 = torch.randn(1024, device=DEV)
w2 = torch.randn(256, 1024, device=DEV) / 32; b2 = torch.randn(256, device=DEV)
wo = torch.randn(256, 256, device=DEV) / 16
g, be = torch.randn(256, device=DEV), torch.randn(256, device=DEV)
h = torch.relu(torch.randn(SN, 1024, device=DEV))
voa_in = torch.empty(SN, 544, device=DEV)
hip.linear(x, wvoa, bvoa, voa_in, aadd=pos, aadd_from_col=256)
jobs = {}
jobs["voa (k_gemm_tile + addend)"] = (lambda o: hip.linear(x, wvoa, bvoa, o, aadd=pos, aadd_from_col=256), (SN, 544))
jobs["ffn1 (k_gemm_tile relu)"] = (lambda o: hip.linear(x, w1, b1, o, relu=True), (SN, 1024))
jobs["msda"] = (lambda o: hip.msda(voa_in, 544, voa_in.view(-1)[256:], 544, o, 1, shapes), (1, SN, 256))
jobs["msda low occupancy (84 VGPRs)"] = (lambda o: hip.msda(voa_in, 544, voa_in.view(-1)[256:], 544, o, 1, shapes, flags=4), (1, SN, 256))
jobs["msda persistent"] = (lambda o: hip.msda(voa_in, 544, voa_in.view(-1)[256:], 544, o, 1, shapes, flags=1), (1, SN, 256))
NL = 1 << 18
glines = torch.randn(NL * 32, device=DEV)
gidx = torch.randint(0, NL, (16384, 32, 12), device=DEV, dtype=torch.int32)
jobs["gather probe (loads only)"] = (lambda o: hip.gather_probe(glines, gidx, o, 16384, NL - 1), (16384, 256))
jobs["rowln K=256"] = (lambda o: hip.linear_res_ln(x, wo, b2, pos, g, be, o), (SN, 256))
jobs["rowln K=1024"] = (lambda o: hip.linear_res_ln(h, w2, b2, x, g, be, o), (SN, 256))
jobs["layernorm"] = (lambda o: hip.layernorm(x, g, be, o), (SN, 256))
M, N, K = 21950, 1024, 256
xs, ws = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / 16
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
hammer_out = torch.empty(M, N, device=DEV)
with hip.split_gemm(False):
    for name, (job, shp) in jobs.items():
        want = torch.zeros(*shp, device=DEV); job(want); torch.cuda.synchronize()
        for hammer in ("split", "default"):
            outs = [torch.zeros(*shp, device=DEV) for _ in range(30)]
            torch.cuda.synchronize()
            with torch.cuda.stream(s1):
                for _ in range(120):
                    hip.gemm(xs, ws, hammer_out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, split=(hammer == "split"))
            with torch.cuda.stream(s2):
                for o in outs:
                    job(o)
            torch.cuda.synchronize()
            bad = [o for o in outs if not torch.equal(o, want)]
            msg = ""
            if bad:
                d = (bad[0] - want).abs(); nz = d.nonzero()
                msg = " first: %d wrong elements, max %.3e, index range %s .. %s" % (len(nz), d.max().item(), nz.min(0).values.tolist(), nz.max(0).values.tolist())
            print("%-30s beside %-7s GEMM loop: wrong %d / 30%s" % (name, hammer, len(bad), msg), flush=True)
