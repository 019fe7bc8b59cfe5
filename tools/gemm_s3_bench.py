"""Stand-alone rates of pn_gemm_s3_f32 on the encoder's GEMM shapes beside the fp32-MFMA kernels
they replace (HIP events over 30 launches, N(0,1) data).  python tools/gemm_s3_bench.py"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pairnet_amd import hip

DEV = "cuda:0"


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def s3(x, add=None):
    out = torch.empty(hip.s3_floats(*x.shape), device=DEV)
    hip.s3_split(x, out, add=add)
    return out


def main():
    hip.lib()
    torch.manual_seed(0)
    for (M, N, K, relu, ln) in [(21950, 1024, 256, True, False), (21950, 256, 1024, False, True),
                                (21950, 544, 256, False, False), (21950, 256, 256, False, True),
                                (66800, 256, 256, False, False), (16700, 512, 256, False, False)]:
        a, w, b = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / 16, torch.randn(N, device=DEV)
        res, g, be = torch.randn(M, 256, device=DEV), torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
        pos = (hip.pos8(torch.randn(M, 256, device=DEV)), M)
        a_s, w_s, res_s = s3(a), s3(w), s3(res)
        out = torch.empty(M, N, device=DEV)
        out_s = torch.empty(hip.s3_floats(M, N), device=DEV)
        out_p = torch.empty(hip.s3_floats(M, N), device=DEV)
        gf = 2.0 * M * N * K * 1e-6
        rows = []
        if ln:
            rows.append(("fp32 MFMA k_gemm_rowln", timed(lambda: hip.linear_res_ln(a, w, b, res, g, be, out))))
            rows.append(("s3 LN -> fp32", timed(lambda: hip.gemm_s3(a_s, w_s, M, N, K, bias=b, out=out, res_s3=res_s, gamma=g, beta=be))))
            rows.append(("s3 LN -> fp32 + S3 + S3(+pos)", timed(lambda: hip.gemm_s3(
                a_s, w_s, M, N, K, bias=b, out=out, out_s3=out_s, out_s3_pos=out_p, pos=pos, res_s3=res_s, gamma=g, beta=be))))
            rows.append(("s3 LN -> S3", timed(lambda: hip.gemm_s3(a_s, w_s, M, N, K, bias=b, out_s3=out_s, res_s3=res_s, gamma=g, beta=be))))
        else:
            rows.append(("fp32 MFMA k_gemm_tile", timed(lambda: hip.linear(a, w, b, out, relu=relu))))
            rows.append(("s3 -> fp32", timed(lambda: hip.gemm_s3(a_s, w_s, M, N, K, bias=b, relu=relu, out=out))))
            rows.append(("s3 -> S3", timed(lambda: hip.gemm_s3(a_s, w_s, M, N, K, bias=b, relu=relu, out_s3=out_s))))
            if N >= 512:
                rows.append(("s3 -> S3, 96-row tile", timed(lambda: hip.gemm_s3(a_s, w_s, M, N, K, bias=b, relu=relu, out_s3=out_s, tile96=True))))
                rows.append(("s3 -> fp32, 96-row tile", timed(lambda: hip.gemm_s3(a_s, w_s, M, N, K, bias=b, relu=relu, out=out, tile96=True))))
        rows.append(("split A", timed(lambda: hip.s3_split(a, a_s))))
        print("M=%d N=%d K=%d" % (M, N, K))
        for name, us in rows:
            print("  %-34s %8.1f us  %6.1f TFLOP/s fp32-eq (%.2f of 157.3)" % (name, us, gf / us, gf / us / 157.3))


if __name__ == "__main__":
    main()
