"""GPU: every C-ABI kernel against a plain torch fp32 CPU statement of the same op
(floating point: tolerances stated per test) or bit-exactly (index / integer work)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import golden
from oracle import layers as L
from oracle.matrix_learner import MatrixLearnerTiny

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip(built_lib):
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from pairnet_amd import hip as h
    h.lib()
    return h


def R(*shape, seed=0, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed + int(np.prod(shape)) % 9973)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def close(got, ref, tol, what=""):
    got = got.detach().cpu()
    err = (got.double() - ref.double()).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert math.isfinite(err) and err <= tol * scale, "%s max|err| %.3e (scale %.2f)" % (what, err, scale)


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,K", [(21950, 256), (21950, 1024), (4099, 512), (33, 32), (2048, 256)])
def test_linear_residual_layernorm_is_bitwise_the_unfused_pair(hip, M, K):
    """pn_linear_res_ln_f32 (one row-owning launch) against pn_gemm_f32 with its residual
    epilogue followed by pn_layernorm_f32: same products in the same order, same LayerNorm
    arithmetic -- bit for bit, including a ragged last row tile; and against torch on the CPU
    within fp32 re-association (2e-5 of the output scale)."""
    x, w = R(M, K, seed=1).to(DEV), (R(256, K, seed=2) * 0.2).to(DEV)
    b, res = R(256, seed=3).to(DEV), R(M, 256, seed=4).to(DEV)
    g, be = R(256, seed=5, lo=0.5, hi=1.5).to(DEV), R(256, seed=6).to(DEV)
    pre, want = torch.empty(M, 256, device=DEV), torch.empty(M, 256, device=DEV)
    hip.linear(x, w, b, pre, res=res, force="tile64")
    hip.layernorm(pre, g, be, want)
    got = torch.full((M, 256), float("nan"), device=DEV)
    hip.linear_res_ln(x, w, b, res, g, be, got)
    assert torch.equal(got, want)
    ref = F.layer_norm(res.cpu().double() + x.cpu().double() @ w.cpu().double().T + b.cpu().double(),
                       (256,), g.cpu().double(), be.cpu().double(), 1e-5)
    close(got, ref, 2e-5, "linear+res+LN")
    # strided operands: a column window of a wider buffer (the [value | offsets | logits] rows)
    wide = torch.zeros(M, K + 64, device=DEV)
    wide[:, 32:32 + K] = x
    got2 = torch.empty(M, 256, device=DEV)
    hip.linear_res_ln(wide[:, 32:32 + K], w, None, res, g, be, got2)
    hip.linear(x, w, None, pre, res=res, force="tile64")
    hip.layernorm(pre, g, be, want)
    assert torch.equal(got2, want)


def test_linear_residual_layernorm_refuses_what_it_cannot_run(hip):
    x, w = torch.zeros(64, 48, device=DEV), torch.zeros(256, 48, device=DEV)
    v = torch.zeros(256, device=DEV)
    o = torch.zeros(64, 256, device=DEV)
    with pytest.raises(RuntimeError):          # K % 32 != 0
        hip.linear_res_ln(x, w, v, o, v, v, torch.empty_like(o))
    with pytest.raises(AssertionError):        # N != 256
        hip.linear_res_ln(torch.zeros(64, 64, device=DEV), torch.zeros(128, 64, device=DEV), v,
                          torch.zeros(64, 128, device=DEV), v, v, torch.zeros(64, 128, device=DEV))


@pytest.mark.parametrize("force", ["tile", "tile64", "tile128x64", "skinny", None])
@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (100, 134, 256), (1000, 544, 260),
                                   (37, 56, 2048), (129, 64, 96), (5, 33, 8)])
def test_gemm_linear_variants(hip, M, N, K, force):
    x, w, b = R(M, K, seed=1), R(N, K, seed=2), R(N, seed=3)
    pos, res = R(50, K, seed=4), R(M, N, seed=5)
    ref = F.relu(F.linear(x + pos.repeat((M + 49) // 50, 1)[:M], w, b)) + res
    out = torch.empty(M, N, device=DEV)
    hip.linear(x.to(DEV), w.to(DEV), b.to(DEV), out, aadd=pos.to(DEV), res=res.to(DEV),
               relu=True, force=force)
    close(out, ref, 2e-5 * math.sqrt(K / 256), "gemm %s" % force)
    out2 = torch.empty(M, N, device=DEV)
    hip.linear(x.to(DEV), w.to(DEV), None, out2, force=force)
    close(out2, F.linear(x, w), 2e-5 * math.sqrt(K / 256), "gemm plain %s" % force)


@pytest.mark.parametrize("force", ["tile", "tile64", "tile128x64", "skinny"])
def test_gemm_batched_strided_and_colmajor(hip, force):
    B, M, N, K = 3, 210, 256, 64          # M % 4 != 0: scalar column-major path
    a = R(B, K, M, seed=7)                # NCHW-like: [b][k][m]
    w, bias = R(N, K, seed=8), R(N, seed=9)
    ref = torch.einsum("bkm,nk->bmn", a, w) + bias
    out = torch.empty(B, M, N, device=DEV)
    hip.gemm(a.to(DEV), w.to(DEV), out, M=M, N=N, K=K, lda=M, ldw=K, ldc=N, bias=bias.to(DEV),
             batch=B, sA=K * M, sC=M * N, colmajor=True, force=force)
    close(out, ref, 2e-5, "colmajor scalar")
    M2 = 400                               # vector path
    a2 = R(B, K, M2, seed=10)
    out = torch.empty(B, M2, N, device=DEV)
    hip.gemm(a2.to(DEV), w.to(DEV), out, M=M2, N=N, K=K, lda=M2, ldw=K, ldc=N, batch=B,
             sA=K * M2, sC=M2 * N, colmajor=True, force=force)
    close(out, torch.einsum("bkm,nk->bmn", a2, w), 2e-5, "colmajor vec")
    # ragged K (K % 32 != 0) on both column-major paths
    for Mr in (210, 400):
        Kr = 72
        ar, wr = R(B, Kr, Mr, seed=13), R(N, Kr, seed=14)
        out = torch.empty(B, Mr, N, device=DEV)
        hip.gemm(ar.to(DEV), wr.to(DEV), out, M=Mr, N=N, K=Kr, lda=Mr, ldw=Kr, ldc=N, batch=B,
                 sA=Kr * Mr, sC=Mr * N, colmajor=True, force=force)
        close(out, torch.einsum("bkm,nk->bmn", ar, wr), 2e-5, "colmajor ragged K")
    # per-batch W (mask logits: W = mask feature of image b), output with a wide ldc
    q, mf = R(B, 100, 256, seed=11), R(B, 530, 256, seed=12)
    out = torch.empty(B, 100, 530, device=DEV)
    hip.gemm(q.to(DEV), mf.to(DEV), out, M=100, N=530, K=256, lda=256, ldw=256, ldc=530,
             batch=B, sA=100 * 256, sW=530 * 256, sC=100 * 530, force=force)
    close(out, torch.einsum("bqc,bpc->bqp", q, mf), 2e-5, "batched W")


@pytest.mark.parametrize("colmajor", [False, True])
def test_gemm_splitk_matches_single_pass(hip, colmajor):
    """Split-K (few tiles, long K; scratch supplied) == one-pass result up to fp32
    re-association, is deterministic, and keeps the fused epilogue."""
    B, M, N, K = 2, 1050, 256, 2048
    a = R(B, K, M, seed=1) if colmajor else R(B, M, K, seed=1)
    w, bias, res = R(N, K, seed=2), R(N, seed=3), R(B, M, N, seed=4)
    ref = F.relu(F.relu((torch.einsum("bkm,nk->bmn", a, w) if colmajor
                         else torch.einsum("bmk,nk->bmn", a, w)) + bias) + res)
    kw = dict(M=M, N=N, K=K, lda=M if colmajor else K, ldw=K, ldc=N, bias=bias.to(DEV),
              res=res.to(DEV), ldres=N, sRes=M * N, batch=B, sA=K * M, sC=M * N, relu=True,
              relu_after=True, colmajor=colmajor)
    scratch = torch.empty(B * 16 * M * N, device=DEV)
    outs = []
    for sc in (scratch, scratch, None):
        out = torch.empty(B, M, N, device=DEV)
        hip.gemm(a.to(DEV), w.to(DEV), out, scratch=sc, force=None if sc is not None else "tile64",
                 **kw)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])                      # deterministic
    close(outs[0], ref, 2e-5 * math.sqrt(K / 256), "split-K")
    close(outs[2], ref, 2e-5 * math.sqrt(K / 256), "single pass")
    assert float((outs[0] - outs[2]).abs().max()) > 0         # the split path really ran


def test_gemm_group_and_column_split_add(hip):
    """Grouped launch (the decoder K/V projections) == the same problems one by one;
    Aadd restricted to columns >= 256 (the fused [value | offsets | logits] GEMM)."""
    probs, refs = [], []
    for i, (M, B) in enumerate(((1050, 2), (4200, 1), (333, 3), (16700, 1))):
        a, w, b = R(B, M + 7, 256, seed=i), R(256, 256, seed=10 + i), R(256, seed=20 + i)
        pos = R(M if i % 2 else 1, 256, seed=30 + i)
        out = torch.empty(B, M, 256, device=DEV)
        probs.append(dict(A=a.to(DEV), W=w.to(DEV), C=out, M=M, N=256, K=256, lda=256, ldw=256,
                          ldc=256, batch=B, sA=(M + 7) * 256, sC=M * 256, bias=b.to(DEV),
                          aadd=pos.to(DEV), ldaadd=256, aadd_rows=pos.shape[0]))
        refs.append(torch.nn.functional.linear(a[:, :M] + pos, w, b))
    hip.gemm_group(probs)
    for pr, ref in zip(probs, refs):
        close(pr["C"], ref, 2e-5, "group")
        single = torch.empty_like(pr["C"])
        kw = {k: v for k, v in pr.items() if k not in ("A", "W", "C")}
        hip.gemm(pr["A"], pr["W"], single, force="tile", **kw)
        assert torch.equal(single, pr["C"])            # same kernel body, same order
    # the full group (18 problems = the K / V projections of all nine decoder layers) in one
    # launch, and one more is refused
    probs, refs = [], []
    for i in range(hip.GEMM_GROUP_MAX):
        M = (97, 410, 1650)[i % 3]
        a, w = R(1, M, 256, seed=40 + i), R(256, 256, seed=70 + i)
        out = torch.empty(1, M, 256, device=DEV)
        probs.append(dict(A=a.to(DEV), W=w.to(DEV), C=out, M=M, N=256, K=256, lda=256, ldw=256,
                          ldc=256, batch=1, sA=M * 256, sC=M * 256))
        refs.append(torch.nn.functional.linear(a, w))
    hip.gemm_group(probs)
    for pr, ref in zip(probs, refs):
        close(pr["C"], ref, 2e-5, "group of 18")
    with pytest.raises(RuntimeError):
        hip.gemm_group(probs + probs[:1])
    M, SN = 2 * 500, 500
    x, w, b, pos = R(M, 256, seed=1), R(544, 256, seed=2), R(544, seed=3), R(SN, 256, seed=4)
    out = torch.empty(M, 544, device=DEV)
    hip.gemm(x.to(DEV), w.to(DEV), out, M=M, N=544, K=256, lda=256, ldw=256, ldc=544,
             bias=b.to(DEV), aadd=pos.to(DEV), ldaadd=256, aadd_rows=SN, aadd_from_col=256,
             force="tile")
    F_ = torch.nn.functional
    ref = torch.cat([F_.linear(x, w[:256], b[:256]),
                     F_.linear(x + pos.repeat(2, 1), w[256:], b[256:])], -1)
    close(out, ref, 2e-5, "column-split add")


def test_gemm_is_an_fmaf_chain(hip):
    """The f32 MFMA is exact fp32: with integer-valued operands the result is exact."""
    x = torch.randint(-8, 9, (256, 512)).float()
    w = torch.randint(-8, 9, (256, 512)).float()
    for force in ("tile", "skinny"):
        out = torch.empty(256, 256, device=DEV)
        hip.linear(x.to(DEV), w.to(DEV), None, out, force=force)
        assert torch.equal(out.cpu(), x @ w.t())


@pytest.mark.parametrize("M", [1, 3, 4, 5, 33, 35, 36, 37, 64, 65, 68, 69, 96, 97, 100, 101, 132])
def test_gemm_ragged_edges_are_exact(hip, M):
    """Tiles that hang over the problem in M (1 .. 4 valid rows in a wave block, whole blocks
    past M) and in N (200 = 3 x 64 + 8): with integer-valued operands the result must be the
    exact product -- bias, ReLU, residual, batches with their own W, split-K partials.  (Written
    for the round-4 experiment that ran such blocks as 4-row v_mfma_f32_4x4x1 strips / skipped
    them, LABNOTES.md 6.0-r4; kept as the edge-case test of the kernel that stayed.)"""
    g = torch.Generator().manual_seed(M)
    B, N, K = 2, 200, 96
    x = torch.randint(-8, 9, (B, M, K), generator=g).float()
    w = torch.randint(-8, 9, (B, N, K), generator=g).float()
    bias = torch.randint(-8, 9, (N,), generator=g).float()
    res = torch.randint(-8, 9, (B, M, N), generator=g).float()
    want = F.relu(torch.einsum("bmk,bnk->bmn", x, w) + bias) + res
    out = torch.full((B, M, N), float("nan"), device=DEV)
    hip.gemm(x.to(DEV), w.to(DEV), out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias.to(DEV),
             res=res.to(DEV), ldres=N, sRes=M * N, relu=True, batch=B, sA=M * K, sW=N * K,
             sC=M * N, force="tile64")
    assert torch.equal(out.cpu(), want), M
    # split-K partials of a ragged block (K = 2048: scratch supplied), single problem
    K2 = 2048
    x2 = torch.randint(-4, 5, (M, K2), generator=g).float()
    w2 = torch.randint(-4, 5, (N, K2), generator=g).float()
    out2 = torch.full((M, N), float("nan"), device=DEV)
    hip.linear(x2.to(DEV), w2.to(DEV), bias.to(DEV), out2, force="tile64",
               scratch=torch.empty(8 * 1024 * 1024, device=DEV))
    assert torch.equal(out2.cpu(), x2 @ w2.t() + bias), M


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,relu", [(2, 13, 17, 256, 256, 3, False),
                                                  (1, 20, 20, 64, 64, 7, True),
                                                  (2, 9, 40, 32, 64, 7, True)])
def test_conv2d_nhwc(hip, B, H, W, Cin, Cout, k, relu):
    x, w, b = R(B, Cin, H, W, seed=1), R(Cout, Cin, k, k, seed=2, lo=-0.05, hi=0.05), R(Cout, seed=3)
    ref = F.conv2d(x, w, b, padding=k // 2)
    ref = F.relu(ref) if relu else ref
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
    out = torch.empty(B, H, W, Cout, device=DEV)
    hip.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV), wp.to(DEV), b.to(DEV), out, B, H,
                    W, Cin, Cout, k, k, k // 2, relu)
    close(out.permute(0, 3, 1, 2), ref, 3e-5, "conv%dx%d" % (k, k))


# ----------------------------------------------------------------------------- norms
def test_layernorm_l2norm(hip):
    x, g, b = R(1003, 256, seed=1, lo=-3, hi=5), R(256, seed=2), R(256, seed=3)
    out = torch.empty(1003, 256, device=DEV)
    hip.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), out)
    close(out, F.layer_norm(x, (256,), g, b, 1e-5), 2e-6, "layernorm")
    hip.l2normalize(x.to(DEV), out)
    close(out, F.normalize(x, p=2, dim=-1, eps=1e-12), 1e-6, "l2norm")


@pytest.mark.parametrize("B,H,W,Cin,Cout,relu", [(1, 24, 32, 256, 256, False),
                                                 (2, 10, 14, 64, 128, True), (1, 2, 2, 32, 32, False),
                                                 (2, 25, 42, 128, 128, True), (1, 13, 21, 64, 32, False),
                                                 (1, 3, 3, 32, 32, True)])
def test_conv3x3_winograd_matches_torch(hip, B, H, W, Cin, Cout, relu):
    x, w, b = R(B, Cin, H, W, seed=1), R(Cout, Cin, 3, 3, seed=2) * 0.05, R(Cout, seed=3)
    ref = F.conv2d(x, w, b, padding=1)
    if relu:
        ref = F.relu(ref)
    T = B * ((H + 1) // 2) * ((W + 1) // 2)    # (odd sides: the last tiles stick out)
    V, Mb = torch.empty(16, T, Cin, device=DEV), torch.empty(16, T, Cout, device=DEV)
    out = torch.full((B, H, W, Cout), float("nan"), device=DEV)
    hip.conv3x3_winograd(x.permute(0, 2, 3, 1).contiguous().to(DEV), hip.winograd_weights(w).to(DEV),
                         b.to(DEV), out, V, Mb, B, H, W, Cin, Cout, relu)
    close(out.permute(0, 3, 1, 2), ref, 3e-5 * math.sqrt(9 * Cin / 256), "winograd 3x3")


@pytest.mark.parametrize("B,H,W,Cin,Cout,relu", [(1, 24, 32, 256, 256, False),
                                                 (2, 10, 15, 64, 128, True), (1, 3, 5, 32, 32, False)])
def test_conv3x3_winograd43_matches_torch(hip, B, H, W, Cin, Cout, relu):
    x, w, b = R(B, Cin, H, W, seed=1), R(Cout, Cin, 3, 3, seed=2) * 0.05, R(Cout, seed=3)
    ref = F.conv2d(x, w, b, padding=1)
    if relu:
        ref = F.relu(ref)
    T = B * ((H + 3) // 4) * ((W + 3) // 4)
    V, Mb = torch.empty(36, T, Cin, device=DEV), torch.empty(36, T, Cout, device=DEV)
    out = torch.empty(B, H, W, Cout, device=DEV)
    hip.conv3x3_winograd43(x.permute(0, 2, 3, 1).contiguous().to(DEV),
                           hip.winograd43_weights(w).to(DEV), b.to(DEV), out, V, Mb, B, H, W, Cin,
                           Cout, relu)
    close(out.permute(0, 3, 1, 2), ref, 2e-4 * math.sqrt(9 * Cin / 256), "winograd F(4,3)")


@pytest.mark.parametrize("M,hidden", [(100, 2048), (200, 2048), (37, 128)])
def test_fused_ffn_layernorm(hip, M, hidden):
    x = R(M, 256, seed=1, lo=-2, hi=2)
    W1, b1 = R(hidden, 256, seed=2, lo=-0.1, hi=0.1), R(hidden, seed=3)
    W2, b2 = R(256, hidden, seed=4, lo=-0.05, hi=0.05), R(256, seed=5)
    g, b = R(256, seed=6), R(256, seed=7)
    ref = F.layer_norm(x + F.linear(F.relu(F.linear(x, W1, b1)), W2, b2), (256,), g, b, 1e-5)
    out = torch.empty(M, 256, device=DEV)
    scr = torch.empty(hip.ffn_scratch_floats(M, hidden), device=DEV)
    hip.ffn_ln(x.to(DEV), W1.to(DEV), b1.to(DEV), W2.to(DEV), b2.to(DEV), g.to(DEV), b.to(DEV),
               out, scr, M, hidden)
    close(out, ref, 5e-6, "ffn+ln")
    # second LayerNorm output (the decoder's post_norm): bit for bit pn_layernorm_f32 of `out`
    g2, b2n = R(256, seed=8).to(DEV), R(256, seed=9).to(DEV)
    out_b, y2 = torch.empty(M, 256, device=DEV), torch.empty(M, 256, device=DEV)
    hip.ffn_ln(x.to(DEV), W1.to(DEV), b1.to(DEV), W2.to(DEV), b2.to(DEV), g.to(DEV), b.to(DEV),
               out_b, scr, M, hidden, post=(g2, b2n, y2))
    want2 = torch.empty(M, 256, device=DEV)
    hip.layernorm(out, g2, b2n, want2)
    assert torch.equal(out_b, out) and torch.equal(y2, want2)


@pytest.mark.parametrize("relu", [False, True])
def test_groupnorm_nhwc_with_batch_strides(hip, relu):
    B, HW = 2, 1050
    x = R(B, 256, HW, seed=4, lo=-2, hi=6)
    g, b = R(256, seed=5), R(256, seed=6)
    ref = F.group_norm(x, 32, g, b, 1e-5)
    ref = F.relu(ref) if relu else ref
    xin = x.permute(0, 2, 1).contiguous().to(DEV)
    big = torch.zeros(B, HW + 77, 256, device=DEV)      # output embedded in a token buffer
    part = torch.empty(B * hip.groupnorm_nblk(HW) * 64, device=DEV, dtype=torch.float64)
    hip.groupnorm_nhwc(xin, g.to(DEV), b.to(DEV), big[:, 77:], part, B, HW, 32, relu, HW * 256,
                       (HW + 77) * 256)
    close(big[:, 77:].permute(0, 2, 1), ref, 3e-6, "groupnorm")
    assert float(big[:, :77].abs().max()) == 0.0


# ----------------------------------------------------------------------------- MSDA
def _msda_ref(value, off, logits, shapes):
    bs, n = value.shape[:2]
    refs = []
    for h, w in shapes:
        yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5,
                                torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
        refs.append(torch.stack([xx.reshape(-1) / w, yy.reshape(-1) / h], -1))
    ref = torch.cat(refs, 0)[None, :, None].repeat(bs, 1, len(shapes), 1)
    norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)
    loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    aw = logits.softmax(-1).view(bs, n, 8, len(shapes), 4)
    return L.msda_core(value, shapes, loc, aw)


def _run_msda(hip, value, off, logits, shapes):
    bs, n = value.shape[:2]
    offaw = torch.cat([off.reshape(bs, n, -1), logits.reshape(bs, n, -1)], -1).contiguous()
    out = torch.empty(bs, n, 256, device=DEV)
    hip.msda(value.reshape(bs, n, 256).to(DEV), 256, offaw.to(DEV), offaw.shape[-1], out, bs, shapes)
    # same data interleaved as [value | offsets | logits] rows (the fused-GEMM layout)
    voa = torch.cat([value.reshape(bs, n, 256), offaw], -1).contiguous().to(DEV)
    out2 = torch.empty(bs, n, 256, device=DEV)
    hip.msda(voa, voa.shape[-1], voa.view(-1)[256:], voa.shape[-1], out2, bs, shapes)
    assert torch.equal(out, out2)
    # the persistent, software-pipelined launch forms (round 4: measured slower, kept as the
    # A/B) compute the same arithmetic in the same order as the default: bit-identical
    for flags in (hip.MSDA_PERSISTENT, hip.MSDA_PERSISTENT_BATCHED, hip.MSDA_LOW_OCCUPANCY):
        out3 = torch.full((bs, n, 256), float("nan"), device=DEV)
        hip.msda(voa, voa.shape[-1], voa.view(-1)[256:], voa.shape[-1], out3, bs, shapes,
                 flags=flags)
        assert torch.equal(out, out3)
    return out


def test_msda_golden_and_random(hip):
    fx = golden("msda")
    shapes = [tuple(s) for s in fx["shapes"].tolist()]
    value, off, logits = (torch.from_numpy(fx[k]) for k in ("value", "offsets", "logits"))
    out = _run_msda(hip, value, off, logits, shapes)
    close(out, torch.from_numpy(fx["out"]), 2e-6, "msda golden")
    shapes = [(7, 11), (13, 21), (25, 42)]
    n = sum(h * w for h, w in shapes)
    value, off = R(1, n, 8, 32, seed=1), R(1, n, 8, 3, 4, 2, seed=2, lo=-6, hi=6)
    logits = R(1, n, 8, 12, seed=3, lo=-3, hi=3)
    close(_run_msda(hip, value, off, logits, shapes), _msda_ref(value, off, logits, shapes), 2e-6,
          "msda random")


@pytest.mark.parametrize("shapes,bs", [([(5, 7)], 2), ([(3, 5), (6, 9)], 1),
                                        ([(2, 3), (4, 5), (7, 9), (13, 17)], 2),
                                        ([(9, 1), (17, 3), (33, 5)], 1)])
def test_msda_level_counts_batches_and_odd_bands(hip, shapes, bs):
    """1 / 2 / 4 levels, batch > 1, maps whose per-band token counts are odd (the second
    query slot of a workgroup is empty) and offsets that leave the map on every side."""
    nl = len(shapes)
    n = sum(h * w for h, w in shapes)
    value = R(bs, n, 8, 32, seed=11)
    off = R(bs, n, 8, nl, 4, 2, seed=12, lo=-9, hi=9)
    logits = R(bs, n, 8, nl * 4, seed=13, lo=-3, hi=3)
    close(_run_msda(hip, value, off, logits, shapes), _msda_ref(value, off, logits, shapes), 2e-6,
          "msda %d levels" % nl)


def _mmcv_inputs(shapes, bs, nq, seed, lo=-0.3, hi=1.3):
    n = sum(h * w for h, w in shapes)
    nl = len(shapes)
    value = R(bs, n, 8, 32, seed=seed)
    loc = R(bs, nq, 8, nl, 4, 2, seed=seed + 1, lo=lo, hi=hi)        # some samples off the map
    aw = R(bs, nq, 8, nl * 4, seed=seed + 2, lo=-3, hi=3).softmax(-1).view(bs, nq, 8, nl, 4)
    ss = torch.tensor(shapes, dtype=torch.int64)
    starts = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    return value, ss, starts, loc, aw


@pytest.mark.parametrize("shapes,bs,nq", [([(3, 4), (6, 8), (12, 16)], 2, 252),
                                           ([(5, 7)], 1, 1), ([(2, 3), (4, 5), (7, 9), (13, 17)], 2, 301),
                                           ([(9, 1), (17, 3)], 3, 100)])
def test_mmcv_shaped_msda_operator(hip, shapes, bs, nq):
    """`ms_deform_attn_forward` in mmcv's own signature (explicit sampling locations and
    attention weights, any number of queries, level geometry from device tensors) against
    the oracle's restatement of mmcv's formula."""
    from pairnet_amd.mmcv_ops import ms_deform_attn_forward
    value, ss, starts, loc, aw = _mmcv_inputs(shapes, bs, nq, 21)
    out = ms_deform_attn_forward(value.to(DEV), ss.to(DEV), starts.to(DEV), loc.to(DEV),
                                 aw.to(DEV), 64)
    assert out.shape == (bs, nq, 256)
    close(out, L.msda_core(value, shapes, loc, aw), 2e-6, "mmcv-shaped msda")


def test_mmcv_shaped_msda_on_the_golden_fixture_equals_the_fused_entry(hip):
    """G6 through the mmcv-shaped entry: locations / weights prepared the way
    MultiScaleDeformableAttention.forward prepares them; same values as the fused encoder
    entry (pn_msda_f32) bit for bit, and the recorded output."""
    from pairnet_amd.mmcv_ops import ms_deform_attn_forward
    fx = golden("msda")
    shapes = [tuple(s) for s in fx["shapes"].tolist()]
    value, off, logits = (torch.from_numpy(fx[k]) for k in ("value", "offsets", "logits"))
    bs, n = value.shape[:2]
    refs = []
    for h, w in shapes:
        yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5,
                                torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
        refs.append(torch.stack([xx.reshape(-1) / w, yy.reshape(-1) / h], -1))
    ref = torch.cat(refs, 0)[None, :, None].repeat(bs, 1, len(shapes), 1)
    norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)
    loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    aw = logits.softmax(-1).view(bs, n, 8, len(shapes), 4)
    ss = torch.tensor(shapes, dtype=torch.int64)
    starts = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    out = ms_deform_attn_forward(value.to(DEV), ss.to(DEV), starts.to(DEV), loc.to(DEV),
                                 aw.to(DEV))
    close(out, torch.from_numpy(fx["out"]), 2e-6, "msda golden via the mmcv-shaped entry")
    close(out, _run_msda(hip, value, off, logits, shapes).cpu(), 2e-6, "mmcv-shaped vs fused entry")


def _msda_grads_ref(value, shapes, loc, aw, gout):
    """Autograd through the oracle's (HF-pinned) restatement of mmcv's CPU formula, in fp64."""
    v, l, a = (t.double().clone().requires_grad_(True) for t in (value, loc, aw))
    out = L.msda_core(v, shapes, l, a)
    out.backward(gout.double())
    return out.detach().float(), v.grad.float(), l.grad.float(), a.grad.float()


def _rel_close(got, want, tol, what):
    got, want = got.detach().cpu().float(), want.float()
    err = float((got - want).abs().max()) / max(float(want.abs().max()), 1e-30)
    assert err < tol, "%s: max err / max |ref| = %.3e" % (what, err)


@pytest.mark.parametrize("shapes,bs,nq,lo,hi", [([(3, 4), (6, 8), (12, 16)], 2, 252, -0.3, 1.3),
                                                 ([(5, 7)], 1, 1, 0.0, 1.0),
                                                 ([(2, 3), (4, 5), (7, 9), (13, 17)], 2, 301, -0.2, 1.2),
                                                 ([(9, 1), (17, 3)], 3, 100, -1.0, 2.0),
                                                 ([(25, 42), (50, 84), (100, 167)], 1, 21950, -0.05, 1.05)])
def test_mmcv_shaped_msda_backward(hip, shapes, bs, nq, lo, hi):
    """`ms_deform_attn_backward` in mmcv's own signature (the second half of
    the reference's one native boundary) against autograd through the oracle's sampling, 1e-4
    relative: small pyramids with samples off the map on every side, and the production
    pyramid of an 800 x 1333 image with one query per token."""
    from pairnet_amd.mmcv_ops import ms_deform_attn_backward
    value, ss, starts, loc, aw = _mmcv_inputs(shapes, bs, nq, 31, lo=lo, hi=hi)
    gout = R(bs, nq, 256, seed=34)
    _, gv, gl, ga = _msda_grads_ref(value, shapes, loc, aw, gout)
    d = lambda t: t.to(DEV)
    grad_value, grad_loc, grad_aw = (torch.zeros_like(d(t)) for t in (value, loc, aw))
    grad_loc.fill_(7.0)                  # (overwritten, whatever it held)
    ms_deform_attn_backward(d(value), d(ss), d(starts), d(loc), d(aw), d(gout), grad_value,
                            grad_loc, grad_aw, 64)
    _rel_close(grad_value, gv, 1e-4, "grad_value")
    # the gradient with respect to a sampling location is piecewise constant in the location:
    # it jumps where the sample crosses a pixel centre, and fp32 (the kernel, like mmcv's) and
    # fp64 (the reference here) may floor() a coordinate within rounding of an integer to
    # different cells -- a discrete decision, exempted like the mask-bit and top-k near-ties
    hw = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float64)     # (x, y) order
    pix = loc.double() * hw[None, None, None, :, None, :] - 0.5
    near = ((pix - pix.round()).abs() < 1e-3).any(-1, keepdim=True).expand_as(gl)
    assert float(near.float().mean()) < 0.01
    _rel_close(torch.where(near, gl, grad_loc.cpu()), gl, 1e-4, "grad_sampling_loc")
    _rel_close(grad_aw, ga, 1e-4, "grad_attn_weight")
    # grad_value is accumulated into: a second call doubles it
    ms_deform_attn_backward(d(value), d(ss), d(starts), d(loc), d(aw), d(gout), grad_value,
                            grad_loc, grad_aw, 64)
    _rel_close(grad_value, 2 * gv, 1e-4, "grad_value accumulated")
    _rel_close(grad_aw, ga, 1e-4, "grad_attn_weight overwritten")


def test_mmcv_autograd_function_matches_the_oracle(hip):
    """`MultiScaleDeformableAttnFunction.apply` (mmcv's autograd function of that name): the
    forward value and all three gradients through torch.autograd."""
    from pairnet_amd.mmcv_ops import MultiScaleDeformableAttnFunction
    shapes = [(6, 8), (12, 16), (24, 32)]
    value, ss, starts, loc, aw = _mmcv_inputs(shapes, 2, 333, 41)
    gout = R(2, 333, 256, seed=44)
    out_ref, gv, gl, ga = _msda_grads_ref(value, shapes, loc, aw, gout)
    v, l, a = (t.to(DEV).requires_grad_(True) for t in (value, loc, aw))
    out = MultiScaleDeformableAttnFunction.apply(v, ss.to(DEV), starts.to(DEV), l, a, 64)
    out.backward(gout.to(DEV))
    close(out.detach(), out_ref, 2e-6, "autograd function forward")
    _rel_close(v.grad, gv, 1e-4, "value.grad")
    _rel_close(l.grad, gl, 1e-4, "sampling_locations.grad")
    _rel_close(a.grad, ga, 1e-4, "attention_weights.grad")
    with pytest.raises(RuntimeError):
        from pairnet_amd.mmcv_ops import ms_deform_attn_backward
        ms_deform_attn_backward(v.detach(), ss.to(DEV), starts.to(DEV), l.detach(), a.detach(),
                                gout.to(DEV), torch.zeros(1, device=DEV), torch.zeros_like(l),
                                torch.zeros_like(a))


# ----------------------------------------------------------------------------- PE / resize
def test_sine_pe(hip):
    pe = L.SinePositionalEncoding(128, normalize=True)
    for h, w in ((25, 42), (3, 4)):
        ref = pe(torch.zeros(1, h, w, dtype=torch.bool))[0].permute(1, 2, 0).reshape(h * w, 256)
        add = R(256, seed=1)
        out = torch.empty(h * w, 256, device=DEV)
        hip.sine_pe(out, add.to(DEV), h, w)
        close(out, ref + add, 2e-5, "sine pe")   # sinf/powf differ by a few ulp between libms


def test_bilinear(hip):
    x = R(2, 256, 12, 21, seed=1)
    base = R(2, 24, 42, 256, seed=2)
    ref = base.permute(0, 3, 1, 2) + F.interpolate(x, (24, 42), mode="bilinear", align_corners=False)
    out = base.clone().to(DEV)
    hip.bilinear_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV), out, 2, 12, 21, 24, 42, 256, True,
                      12 * 21 * 256, 24 * 42 * 256)
    close(out.permute(0, 3, 1, 2), ref, 2e-6, "bilinear nhwc")
    m = R(7, 40, 67, seed=3, lo=-4, hi=4)
    for size in ((5, 9), (10, 17), (20, 34), (77, 131), (77, 132), (96, 160), (3, 4)):   # wo % 4 == 0: the 4-pixel kernel
        ref = F.interpolate(m[None], size, mode="bilinear", align_corners=False)[0]
        out = torch.empty(7, *size, device=DEV)
        hip.bilinear_planar(m.to(DEV), out, 7, 40, 67, size[0], size[1])
        close(out, ref, 2e-6, "bilinear planar %s" % (size,))
        o8 = torch.empty(7, *size, device=DEV, dtype=torch.uint8)
        hip.bilinear_planar_gt0(m.to(DEV), o8, 7, 40, 67, size[0], size[1])
        mism = (o8.cpu().bool() != (torch.sigmoid(ref) > 0.5)) & (ref.abs() > 1e-5)
        assert int(mism.sum()) == 0


# ----------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, mask, scale):
    B, Qn, _ = q.shape
    qh = q.view(B, Qn, 8, 32).transpose(1, 2) * scale
    kh = k.view(B, -1, 8, 32).transpose(1, 2)
    vh = v.view(B, -1, 8, 32).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    if mask is not None:
        s = s.masked_fill(mask[:, None], float("-inf"))
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, Qn, 256)


@pytest.mark.parametrize("B,Qn,Nk,masked", [(1, 100, 1050, True), (2, 100, 4200, True),
                                            (1, 100, 16700, True), (2, 100, 100, False),
                                            (1, 100, 200, False), (1, 37, 333, True),
                                            (1, 200, 777, True)])
def test_attention_and_mask_pack(hip, B, Qn, Nk, masked):
    q, k, v = R(B, Qn, 256, seed=1, lo=-2, hi=2), R(B, Nk, 256, seed=2), R(B, Nk, 256, seed=3)
    scale = 1 / math.sqrt(32)
    mask = bits = rowall = None
    if masked:
        logits = R(B * Qn, Nk, seed=4)
        logits[3] = -logits[3].abs() - 0.1          # an all-masked row -> un-masked (:300)
        logits[5, : Nk - 1] = -1.0                   # exactly one key survives
        logits[5, Nk - 1] = 1.0
        mask = (logits < 0).view(B, Qn, Nk).clone()
        mask[mask.all(-1)] = False
        bits = torch.empty(B * Qn * ((Nk + 31) // 32), device=DEV, dtype=torch.int32)
        rowall = torch.empty(B * Qn, device=DEV, dtype=torch.int32)
        hip.mask_pack(logits.to(DEV), bits, rowall, B * Qn, Nk)
        ra = rowall.cpu()
        assert int(ra[3]) == 1 and int(ra.sum()) == int((logits < 0).all(-1).sum())
        words = bits.cpu().view(B * Qn, -1).numpy().view(np.uint32)
        unpacked = np.unpackbits(words.view(np.uint8), axis=1, bitorder="little")[:, :Nk]
        assert np.array_equal(unpacked.astype(bool), (logits < 0).numpy())
    scr = torch.empty(hip.attn_scratch_floats(B, Qn, Nk), device=DEV)
    out = torch.empty(B, Qn, 256, device=DEV)
    hip.attention(q.to(DEV), 256, k.to(DEV), 256, v.to(DEV), 256, bits, rowall, out, 256, scr, B, Qn,
                  Nk, scale)
    close(out, _attn_ref(q, k, v, mask, scale), 3e-6, "attention")


def test_attention_strided_qk_matches_nn_multiheadattention(hip):
    """Self-attention as the decoder layer issues it: packed [Q|K] projection buffer with
    ld 512; checked against torch.nn.MultiheadAttention itself."""
    mha = torch.nn.MultiheadAttention(256, 8).eval()
    with torch.no_grad():
        mha.in_proj_bias.copy_(R(768, seed=1))
    x, pos = R(100, 2, 256, seed=2, lo=-2, hi=2), R(100, 1, 256, seed=3)
    with torch.no_grad():
        ref = mha(x + pos, x + pos, x, need_weights=False)[0]      # (Q, B, C)
    W, b = mha.in_proj_weight.detach(), mha.in_proj_bias.detach()
    xb = x.transpose(0, 1).reshape(200, 256).contiguous().to(DEV)
    QK, V = torch.empty(200, 512, device=DEV), torch.empty(200, 256, device=DEV)
    hip.linear(xb, W[:512].to(DEV), b[:512].to(DEV), QK, aadd=pos[:, 0].contiguous().to(DEV))
    hip.linear(xb, W[512:].to(DEV), b[512:].to(DEV), V)
    att, out = torch.empty(200, 256, device=DEV), torch.empty(200, 256, device=DEV)
    scr = torch.empty(hip.attn_scratch_floats(2, 100, 100), device=DEV)
    hip.attention(QK, 512, QK[:, 256:], 512, V, 256, None, None, att, 256, scr, 2, 100, 100,
                  1 / math.sqrt(32))
    hip.linear(att, mha.out_proj.weight.detach().to(DEV), mha.out_proj.bias.detach().to(DEV), out)
    close(out.view(2, 100, 256).transpose(0, 1), ref, 5e-6, "mha")


# ----------------------------------------------------------------------------- PPN
def test_matrix_learner_golden(hip):
    fx = golden("convtiny")
    net = MatrixLearnerTiny().eval()
    from oracle import seeded
    sd = seeded.seeded_state_dict({k: v.shape for k, v in net.state_dict().items()},
                                  int(fx["weight_seed"]))
    x = torch.from_numpy(fx["x"])
    B, S = x.shape[0], x.shape[1]
    w1 = sd["conv_layers.0.0.weight"].reshape(64, 49).contiguous().to(DEV)
    w2 = sd["conv_layers.1.0.weight"].permute(0, 2, 3, 1).reshape(64, -1).contiguous().to(DEV)
    w3 = sd["conv_layers.2.0.weight"].reshape(64, 49).t().contiguous().to(DEV)
    c1, c2 = torch.empty(B, S * S, 64, device=DEV), torch.empty(B, S * S, 64, device=DEV)
    out = torch.empty(B, S, S, device=DEV)
    hip.mlearner_first(x.to(DEV), w1, sd["conv_layers.0.0.bias"].to(DEV), c1, B, S)
    hip.conv2d_nhwc(c1, w2, sd["conv_layers.1.0.bias"].to(DEV), c2, B, S, S, 64, 64, 7, 7, 3, True)
    hip.mlearner_last(c2, w3, sd["conv_layers.2.0.bias"].to(DEV), out, B, S)
    net.load_state_dict(sd)
    with torch.no_grad():
        c1_ref = F.relu(net.conv_layers[0][0](x[:, None]))
    close(c1.view(B, S, S, 64).permute(0, 3, 1, 2), c1_ref, 2e-6, "ml first")
    close(out, torch.from_numpy(fx["y"]), 1e-5, "matrix learner (reference ConvTiny golden)")


def _topk_ref(scores, k):
    """value descending, index ascending on ties."""
    order = np.lexsort((np.arange(scores.size), -scores.astype(np.float64)))
    return order[:k]


@pytest.mark.parametrize("Q,k", [(100, 100), (200, 100), (37, 7), (16, 256), (256, 128)])
def test_topk_pairs_bit_exact(hip, Q, k):
    B = 3
    s = R(B, Q * Q, seed=Q)
    s[1] = torch.round(s[1] * 20) / 20             # massive ties
    s[2, :50] = s[2].max() + 1.0                   # tied maxima at the head
    s[0, 17] = -0.0
    idx = torch.empty(B, k, device=DEV, dtype=torch.int64)
    sub, obj = torch.empty_like(idx), torch.empty_like(idx)
    hip.topk_pairs(s.to(DEV), idx, sub, obj, B, Q, k)
    idx, sub, obj = idx.cpu(), sub.cpu(), obj.cpu()
    for b in range(B):
        assert np.array_equal(idx[b].numpy(), _topk_ref(s[b].numpy(), k)), b
    assert torch.equal(sub, torch.div(idx, Q, rounding_mode="trunc"))
    assert torch.equal(obj, torch.remainder(idx, Q))
    # tie-free row: identical to torch.topk (sorted descending), as the reference calls it
    assert torch.equal(idx[0], torch.topk(s[0], k)[1]) or len(set(s[0].tolist())) < Q * Q


def test_topk_on_reference_importance_is_bit_exact(hip):
    """PPN golden: fed the reference's own importance tensor, the selector reproduces the
    reference's top-k indices, sub_pos and obj_pos bit for bit."""
    fx = golden("ppn")
    imp = torch.from_numpy(fx["importance"])
    idx = torch.empty(1, 100, device=DEV, dtype=torch.int64)
    sub, obj = torch.empty_like(idx), torch.empty_like(idx)
    hip.topk_pairs(imp.to(DEV), idx, sub, obj, 1, 100, 100)
    assert np.array_equal(idx.cpu().numpy(), fx["topk_idx"])
    assert np.array_equal(sub.cpu().numpy(), fx["sub_pos"])
    assert np.array_equal(obj.cpu().numpy(), fx["obj_pos"])


def test_gather_rows(hip):
    B, rin, rout = 2, 100, 64
    for length in (256, 134, 66800):
        x = R(B, rin, length, seed=length)
        index = torch.randint(0, rin, (B, rout))
        out = torch.empty(B, rout, length, device=DEV)
        hip.gather_rows(x.to(DEV), index.to(DEV), out, B, rin, rout, length)
        ref = torch.gather(x, 1, index[..., None].expand(-1, -1, length))
        assert torch.equal(out.cpu(), ref)


# ----------------------------------------------------------------------------- post-processing
def test_cls_argmax_rel_dists_panoptic(hip):
    logits = R(200, 134, seed=1, lo=-4, hi=4)
    lab = torch.empty(200, device=DEV, dtype=torch.int64)
    sc = torch.empty(200, device=DEV)
    hip.cls_argmax(logits.to(DEV), lab, sc, 200, 134)
    p = F.softmax(logits, -1)[..., :-1]
    assert torch.equal(lab.cpu(), p.argmax(-1))
    close(sc, p.max(-1)[0], 1e-6, "score")
    rl = R(100, 56, seed=2, lo=-3, hi=3)
    out = torch.empty(100, 57, device=DEV)
    hip.rel_dists(rl.to(DEV), out, 100, 56)
    close(out, torch.cat([torch.zeros(100, 1), F.softmax(rl, -1)], -1), 1e-6, "rel dists")
    n, HW = 9, 5000
    m = R(n, HW, seed=3, lo=-5, hi=5)
    labels = torch.randint(0, 133, (n,))
    remap = torch.arange(n, dtype=torch.int32)
    remap[4] = 2
    seg = torch.empty(HW, device=DEV, dtype=torch.int64)
    area = torch.zeros(n, device=DEV, dtype=torch.int32)
    hip.panoptic(m.to(DEV), labels.to(DEV), remap.to(DEV), seg, area, n, HW)
    ids = m.t().softmax(-1).argmax(-1)
    ids = remap.long()[ids]
    assert torch.equal(seg.cpu(), ids * 1000 + labels[ids])
    assert torch.equal(area.cpu().long(), torch.bincount(ids, minlength=n))


@pytest.mark.gpu
@pytest.mark.parametrize("B,Q", [(1, 100), (2, 37), (1, 16), (3, 5)])
def test_ppn_front_equals_normalise_cosine_first_layer(hip, B, Q):
    """k_ppn_front (normalised query tiles in LDS -> cosine block on MFMA -> 7x7 1->64 conv +
    ReLU) against F.normalize / matmul / conv2d (pairnet_head.py:325-333,
    cnn_factory.py:22-29) and against the three-launch path it replaces."""
    g = torch.Generator().manual_seed(7 + Q)
    se, oe = (torch.randn(B, Q, 256, generator=g) * 3 for _ in range(2))
    se[0, Q // 2] = 0                                    # a zero row: eps clamp
    w1 = torch.randn(64, 1, 7, 7, generator=g) * 0.2
    b1 = torch.randn(64, generator=g) * 0.1
    raw_ref = F.normalize(se, dim=-1) @ F.normalize(oe, dim=-1).transpose(1, 2)
    c1_ref = F.relu(F.conv2d(raw_ref[:, None], w1, b1, padding=3)).permute(0, 2, 3, 1)
    d = lambda t: t.contiguous().cuda()
    raw = torch.full((B, Q, Q), float("nan"), device="cuda")
    c1 = torch.full((B, Q, Q, 64), float("nan"), device="cuda")
    hip.ppn_front(d(se), d(oe), d(w1.view(64, 49)), d(b1), raw, c1, B, Q)
    assert torch.isfinite(raw).all() and torch.isfinite(c1).all()
    assert (raw.cpu() - raw_ref).abs().max() < 2e-6
    assert (c1.cpu() - c1_ref).abs().max() < 1e-5
    sn, on = torch.empty_like(d(se)), torch.empty_like(d(oe))
    hip.l2normalize(d(se).view(-1, 256), sn.view(-1, 256))
    hip.l2normalize(d(oe).view(-1, 256), on.view(-1, 256))
    raw2, c12 = torch.empty_like(raw), torch.empty_like(c1)
    hip.gemm(sn, on, raw2, M=Q, N=Q, K=256, lda=256, ldw=256, ldc=Q, batch=B, sA=Q * 256,
             sW=Q * 256, sC=Q * Q)
    hip.mlearner_first(raw2, d(w1.view(64, 49)), d(b1), c12, B, Q)
    assert (raw - raw2).abs().max() < 1e-6 and (c1 - c12).abs().max() < 5e-6


# ----------------------------------------------------------------------------- result copy
@pytest.mark.parametrize("nbytes,wgs", [(1, 1), (15, 4), (16, 4), (4099, 2), (1 << 20, 4),
                                        (51_147_680 // 8 + 7, 16)])
def test_copy_stream_to_pinned_host_and_device(hip, nbytes, wgs):
    """pn_copy_stream: device -> pinned host (the D2H of triplet2Result's fields) and device ->
    device, any byte count (16-byte body + byte tail), any workgroup count."""
    g = torch.Generator().manual_seed(nbytes)
    src = torch.randint(0, 256, (nbytes,), generator=g, dtype=torch.uint8)
    d = src.to(DEV)
    host = torch.zeros(nbytes, dtype=torch.uint8, pin_memory=True)
    dev2 = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    hip.copy_stream(d, host, wgs)
    hip.copy_stream(d, dev2, wgs)
    torch.cuda.synchronize()
    assert torch.equal(host, src) and torch.equal(dev2.cpu(), src)
    # typed tensors, and the argument contract
    f = torch.randn(33, 7, generator=g).to(DEV)
    fh = torch.empty(33, 7, pin_memory=True)
    hip.copy_stream(f, fh)
    torch.cuda.synchronize()
    assert torch.equal(fh, f.cpu())
    with pytest.raises(RuntimeError):
        hip.copy_stream(f, torch.empty(33, 7))                 # pageable host memory
    with pytest.raises(RuntimeError):
        hip.copy_stream(f, torch.empty(7, 33, pin_memory=True).t())   # not contiguous
    assert hip.lib().pn_copy_stream(None, None, 16, 4, None) == -1


@pytest.mark.parametrize("n", [1, 7, 8, 9, 4099, 200 * 80 * 112, 2 * 384 * 640 + 3])
def test_mask_bits_round_trip(hip, n):
    """pn_pack_bool_bits (device) against numpy.packbits(bitorder="little"), any length, any
    non-zero byte = set; pn_unpack_bits_host gives back the bool array."""
    g = torch.Generator().manual_seed(n)
    b = torch.rand(n, generator=g) < 0.37
    want = np.packbits(b.numpy(), bitorder="little")
    bits = torch.zeros((n + 7) // 8 + 5, dtype=torch.uint8, device=DEV)
    hip.pack_bool_bits(b.to(DEV), bits)
    assert np.array_equal(bits.cpu().numpy()[:len(want)], want)
    assert not bits[len(want):].any()                       # nothing written past the end
    raw = torch.where(b, torch.randint(1, 256, (n,), generator=g), 0).to(torch.uint8)
    bits.zero_()
    hip.pack_bool_bits(raw.to(DEV), bits)
    assert np.array_equal(bits.cpu().numpy()[:len(want)], want)
    for threads in (1, 3):
        out = torch.ones(n + 3, dtype=torch.bool)
        hip.unpack_bits_host(bits.cpu(), out[:n], threads)
        assert torch.equal(out[:n], b) and bool(out[n:].all())
    with pytest.raises(RuntimeError):
        hip.pack_bool_bits(b.to(DEV), torch.zeros(max(1, (n + 7) // 8 - 1), dtype=torch.uint8,
                                                  device=DEV)[:(n + 7) // 8 - 1])


@pytest.mark.parametrize("hi,wi,ho,wo", [(200, 334, 25, 42), (200, 334, 50, 84),
                                         (200, 334, 100, 167), (24, 32, 3, 4), (19, 26, 5, 7),
                                         (7, 9, 7, 9), (200, 267, 25, 34), (13, 17, 1, 1)])
def test_stencil_rows_and_stencil_mask_pack(hip, hi, wi, ho, wo):
    """pn_bilinear_stencil_rows_f32 gathers the 4 source rows of every output pixel of an
    align_corners=False bilinear resize; logits against those rows, blended by
    pn_mask_pack_stencil, give bit for bit the mask of (full-resolution logits ->
    pn_bilinear_planar_f32 -> pn_mask_pack)."""
    B, Q, C = 2, 100, 256
    g = torch.Generator().manual_seed(hi * wi + ho)
    mf = torch.randn(B, hi * wi, C, generator=g).to(DEV)
    me = torch.randn(B * Q, C, generator=g).to(DEV)
    n = ho * wo
    rows = torch.empty(B, 4 * n, C, device=DEV)
    hip.bilinear_stencil_rows(mf, rows, B, hi, wi, ho, wo, C, hi * wi * C, 4 * n * C)
    # the gather against ATen's index formula
    def taps(o, i_n, o_n):
        src = (torch.arange(o_n, dtype=torch.float32) + 0.5) * (float(i_n) / float(o_n)) - 0.5
        i0 = src.clamp(min=0).floor().long().clamp(max=i_n - 1)
        return i0, (i0 + (i0 < i_n - 1).long())
    y0, y1 = taps(None, hi, ho)
    x0, x1 = taps(None, wi, wo)
    idx = torch.stack([(yy[:, None] * wi + xx[None, :]).reshape(-1)
                       for yy, xx in ((y0, x0), (y0, x1), (y1, x0), (y1, x1))]).reshape(-1)
    assert torch.equal(rows.cpu(), mf.cpu()[:, idx])
    # dense: full-resolution logits -> resize -> pack
    full = torch.empty(B * Q, hi * wi, device=DEV)
    hip.gemm(me, mf, full, M=Q, N=hi * wi, K=C, lda=C, ldw=C, ldc=hi * wi, batch=B, sA=Q * C,
             sW=hi * wi * C, sC=Q * hi * wi, force="tile64")
    small = torch.empty(B * Q, n, device=DEV)
    hip.bilinear_planar(full, small, B * Q, hi, wi, ho, wo)
    nw = (n + 31) // 32
    bits_d = torch.zeros(B * Q * nw, dtype=torch.int32, device=DEV)
    all_d = torch.zeros(B * Q, dtype=torch.int32, device=DEV)
    hip.mask_pack(small, bits_d, all_d, B * Q, n)
    # sparse: logits of the stencil rows -> blend + pack
    l4 = torch.empty(B * Q, 4 * n, device=DEV)
    hip.gemm(me, rows, l4, M=Q, N=4 * n, K=C, lda=C, ldw=C, ldc=4 * n, batch=B, sA=Q * C,
             sW=4 * n * C, sC=Q * 4 * n, force="tile64")
    bits_s = torch.zeros_like(bits_d)
    all_s = torch.ones_like(all_d)
    hip.mask_pack_stencil(l4, bits_s, all_s, B * Q, hi, wi, ho, wo)
    torch.cuda.synchronize()
    assert torch.equal(bits_s, bits_d) and torch.equal(all_s, all_d)
    # round 5: the same two steps as ONE launch (pn_mask_stencil_gemm_f32: blend / threshold /
    # pack in the GEMM epilogue) -- the same bits and flags, starting from garbage outputs
    bits_f = torch.full_like(bits_d, 0x5a5a5a5a)
    all_f = torch.full_like(all_d, 7)
    hip.mask_stencil_gemm(me, rows, bits_f, all_f, B, Q, hi, wi, ho, wo, K=C)
    torch.cuda.synchronize()
    assert torch.equal(bits_f, bits_d) and torch.equal(all_f, all_d)
    # an all-masked row is flagged (a mask embedding whose logits are all negative: the rows
    # are made non-negative and the embedding -1)
    rows_pos, me_neg = rows.abs(), me.clone()
    me_neg[3] = -1.0
    hip.mask_stencil_gemm(me_neg, rows_pos, bits_f, all_f, B, Q, hi, wi, ho, wo, K=C)
    torch.cuda.synchronize()
    assert int(all_f[3]) == 1 and int(all_f[4]) == int(all_d[4]) == 0
    assert bool((bits_f.view(B * Q, nw)[3].cpu().numpy().view("uint32") != 0).any())
    l4[3].fill_(-1.0)
    hip.mask_pack_stencil(l4, bits_s, all_s, B * Q, hi, wi, ho, wo)
    assert int(all_s[3]) == 1


def test_winograd_weight_transform_kernel_equals_the_einsum():
    """`pn_winograd_weights_f32` (device, what the heads / backbones pack with since round 6)
    against the float64 einsum it replaced (still the host-tensor path): both round G g G^T,
    formed in double, to fp32 once."""
    from pairnet_amd import hip
    w = torch.randn(96, 40, 3, 3, generator=torch.Generator().manual_seed(21))
    for fn, n in ((hip.winograd_weights, 16), (hip.winograd43_weights, 36)):
        dev, host = fn(w.to(DEV)), fn(w)
        assert tuple(dev.shape) == tuple(host.shape) == (n, 96, 40)
        diff = (dev.cpu() - host).abs()
        assert float(diff.max()) <= 2e-7 * float(host.abs().max())       # (last-bit differences:
        assert float((diff > 0).float().mean()) < 2e-2                   #  another summation order)


def test_gemm_with_fewer_than_32_contraction_columns():
    """K < 32 at sizes that used to take the 64x64 tile kernel, whose ragged-tail path loads 32
    columns of every W row (past the last row for K < 32: a GPU fault when the buffer ends a
    mapped segment -- found through the backward pass's dW GEMMs): dispatched to the skinny kernel
    since round 6; row-major and column-major A."""
    from pairnet_amd import hip
    g = torch.Generator().manual_seed(23)
    M, N, K = 2048, 1024, 24
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    want = A.double() @ W.double().t()
    out = torch.empty(M, N, device=DEV)
    hip.gemm(A.to(DEV), W.to(DEV), out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N)
    assert float((out.cpu().double() - want).abs().max()) < 1e-4
    At = A.t().contiguous().to(DEV)                      # [K][M]: A read column-major
    out2 = torch.empty(M, N, device=DEV)
    hip.gemm(At, W.to(DEV), out2, M=M, N=N, K=K, lda=M, ldw=K, ldc=N, colmajor=True)
    assert float((out2.cpu().double() - want).abs().max()) < 1e-4
