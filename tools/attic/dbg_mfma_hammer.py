"""A pure-MFMA spinner (tools/mfma_hammer.hip: no memory, no LDS, no barrier) on one stream; on
another, default-path kernels of the library on fixed inputs.  Whose output changes?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pairnet_amd import hip
hip.lib()
ham = ctypes.CDLL(os.path.join(ROOT, "tools", "bin", "libmfma_hammer.so"))
ham.mfma_hammer.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
DEV = "cuda:0"
torch.manual_seed(0)
SN, shapes = 21950, [(25, 42), (50, 84), (100, 167)]
x = torch.randn(SN, 256, device=DEV); pos = torch.randn(SN, 256, device=DEV)
wvoa = torch.randn(544, 256, device=DEV) / 16; bvoa = torch.randn(544, device=DEV) * 0.1
g, be = torch.randn(256, device=DEV), torch.randn(256, device=DEV)
voa_in = torch.empty(SN, 544, device=DEV)
hip.linear(x, wvoa, bvoa, voa_in, aadd=pos, aadd_from_col=256)
q = torch.randn(100, 256, device=DEV); kk = torch.randn(16700, 256, device=DEV); vv = torch.randn(16700, 256, device=DEV)
scr = torch.empty(hip.lib().pn_attn_scratch_floats(1, 100, 16700), device=DEV)
jobs = {
    "msda": (lambda o: hip.msda(voa_in, 544, voa_in.view(-1)[256:], 544, o, 1, shapes), (1, SN, 256)),
    "attention 16700 keys": (lambda o: hip.attention(q, 256, kk, 256, vv, 256, None, None, o, 256, scr, 1, 100, 16700, 0.1767767), (100, 256)),
}
sink = torch.zeros(4, device=DEV)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
kinds = {0: "bf16 MFMA", 3: "bf16 MFMA, ~124 VGPRs", 4: "VALU, ~124 VGPRs"}
for name, (job, shp) in jobs.items():
    want = torch.zeros(*shp, device=DEV); job(want); torch.cuda.synchronize()
    for kind, kname in kinds.items():
        outs = [torch.zeros(*shp, device=DEV) for _ in range(20)]
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            for _ in range(6):
                ham.mfma_hammer(s1.cuda_stream, sink.data_ptr(), kind, 512, 6000)
        with torch.cuda.stream(s2):
            for o in outs:
                job(o)
        torch.cuda.synchronize()
        bad = [o for o in outs if not torch.equal(o, want)]
        msg = ""
        if bad:
            d = (bad[0] - want).abs(); nz = d.nonzero()
            msg = " first: %d wrong elements, max %.3e, columns %d .. %d" % (len(nz), d.max().item(), nz[:, -1].min().item(), nz[:, -1].max().item())
        print("%-22s beside a %-22s spinner: wrong %2d / 20%s" % (name, kname, len(bad), msg), flush=True)
