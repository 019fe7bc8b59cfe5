"""Which kernel's results change when another kernel is resident beside it?  (Round 5 met a
'packed fp32 beside bf16 MFMA' phenomenon, LABNOTES R5.12; this is the reproducer from a clean
checkout: only library kernels, no patch.)

For every (aggressor, victim) pair: the aggressor is launched in a loop on one stream, the victim
-- fixed inputs -- repeatedly on another stream into separate output buffers; every victim output
is compared bit for bit with the same launch run alone.  Aggressors: the bf16-MFMA GEMM
(pn_gemm_s3_f32) and, as the control, the fp32-MFMA GEMM (pn_gemm_f32).
    python tools/coresidency_probe.py [repeats]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pairnet_amd import hip

DEV = "cuda:0"
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def s3(x, add=None):
    out = torch.empty(hip.s3_floats(*x.shape), device=DEV)
    hip.s3_split(x, out, add=add)
    return out


def main():
    hip.lib()
    torch.manual_seed(0)
    M = 21950
    a, w = torch.randn(M, 256, device=DEV), torch.randn(1024, 256, device=DEV) / 16
    a_s, w_s = s3(a), s3(w)
    big = torch.empty(M, 1024, device=DEV)
    big_s = torch.empty(hip.s3_floats(M, 1024), device=DEV)

    def agg_s3():
        hip.gemm_s3(a_s, w_s, M, 1024, 256, relu=True, out_s3=big_s)

    def agg_f32():
        hip.linear(a, w, None, big, relu=True)

    # ---- victims: (name, launch(out), out shape) ----
    shapes = [(25, 42), (50, 84), (100, 167)]
    SN = sum(h * wd for h, wd in shapes)
    voa = torch.randn(1, SN, 544, device=DEV)
    voa[..., 256:448] *= 2.0
    x256 = torch.randn(SN, 256, device=DEV)
    g, be = torch.rand(256, device=DEV) + 0.5, torch.randn(256, device=DEV)
    w2 = torch.randn(256, 256, device=DEV) / 16
    w2_s, x_s = s3(w2), s3(x256)
    res_s = s3(torch.randn(SN, 256, device=DEV))
    victims = [
        ("k_msda (deformable sampling)", lambda o: hip.msda(voa, 544, voa.view(-1)[256:], 544, o, 1, shapes), (1, SN, 256)),
        ("k_layernorm256", lambda o: hip.layernorm(x256, g, be, o), (SN, 256)),
        ("k_gemm_tile fp32 MFMA 21950x256x256", lambda o: hip.linear(x256, w2, g, o), (SN, 256)),
        ("k_gemm_rowln fp32 MFMA + LayerNorm", lambda o: hip.linear_res_ln(x256, w2, g, x256, g, be, o), (SN, 256)),
        ("k_gemm_s3 bf16 MFMA 21950x256x256", lambda o: hip.gemm_s3(x_s, w2_s, SN, 256, 256, bias=g, out=o), (SN, 256)),
        ("k_gemm_s3<ln> bf16 MFMA + LayerNorm", lambda o: hip.gemm_s3(x_s, w2_s, SN, 256, 256, bias=g, out=o, res_s3=res_s, gamma=g, beta=be), (SN, 256)),
        ("k_s3_split", lambda o: hip.s3_split(x256, o, add=x256), (hip.s3_floats(SN, 256),)),
    ]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for vname, vfn, vshape in victims:
        ref = torch.empty(vshape, device=DEV)
        vfn(ref)
        torch.cuda.synchronize()
        alone = torch.empty(vshape, device=DEV)
        vfn(alone)
        torch.cuda.synchronize()
        assert torch.equal(alone, ref), vname + ": not deterministic alone"
        for aname, afn in (("bf16-MFMA GEMM", agg_s3), ("fp32-MFMA GEMM (control)", agg_f32)):
            outs = [torch.full(vshape, float("nan"), device=DEV) for _ in range(REP)]
            torch.cuda.synchronize()
            with torch.cuda.stream(sa):
                for _ in range(REP * 3):
                    afn()
            with torch.cuda.stream(sb):
                for o in outs:
                    vfn(o)
            torch.cuda.synchronize()
            bad = [o for o in outs if not torch.equal(o, ref)]
            worst = max([float((o - ref).abs().max()) for o in bad], default=0.0)
            nel = max([int((o != ref).sum()) for o in bad], default=0)
            print("%-40s beside %-26s: %2d of %d launches differ (max |diff| %.3e, up to %d elements)" % (
                vname, aname, len(bad), REP, worst, nel))


if __name__ == "__main__":
    main()
