"""GPU: the pre-split-operand GEMM (csrc/gemm_s3.hip, pn_gemm_s3_f32) through the C ABI.

fp32 operands are stored as three bf16 planes (exactly: x = x0 + x1 + x2) and contracted with six
bf16 MFMA products into fp32 accumulators.  The tests state what that arithmetic guarantees:
the split is exact; integer-valued problems are exact; on random data the error against an fp64
product is at or below the exact-fp32 MFMA kernel's (pn_gemm_f32) and far inside the path's 1e-3
bar; every output form (fp32 rows, S3, S3 of out + pos) carries the same values; the
residual + LayerNorm row epilogue matches torch.  Tolerances are written in each test."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip(built_lib):
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from pairnet_amd import hip as h
    h.lib()
    return h


def G(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed * 7919 + int(np.prod(shape)) % 9973)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def s3(hip, x, add=None):
    out = torch.empty(hip.s3_floats(x.shape[0], x.shape[1]), device=DEV)
    hip.s3_split(x, out, add=add)
    return out


def joined(hip, s, rows, K):
    out = torch.full((rows, K), float("nan"), device=DEV)
    hip.s3_join(s, out)
    return out


@pytest.mark.parametrize("rows,K", [(1, 16), (31, 32), (96, 256), (100, 256), (21950, 256), (333, 1024)])
def test_split_is_exact_and_join_returns_the_bits(hip, rows, K):
    """x0 + x1 + x2 == x bit for bit for every fp32 value (normal range), ragged row counts
    included; the addend form splits fl(x + add[row % len])."""
    x = G(rows, K, seed=1)
    x.view(-1)[::7] *= 1e-12
    x.view(-1)[::11] *= 1e9
    x.view(-1)[3] = 0.0
    assert torch.equal(joined(hip, s3(hip, x), rows, K), x)
    add = G(max(1, rows // 3), K, seed=2)
    want = x + add[torch.arange(rows, device=DEV) % add.shape[0]]
    assert torch.equal(joined(hip, s3(hip, x, add), rows, K), want)
    # the planes are bf16 roundings of what is left: |x1| <= 2^-8 |x0| ulp-wise, i.e. the first
    # plane alone is x rounded to bf16
    planes = s3(hip, x).view(torch.int16).view(-1, 3, 512)
    first = planes[:, 0].contiguous().view(torch.bfloat16).float()
    rb, kb = (rows + 31) // 32, K // 16
    first = first.view(rb, kb, 2, 32, 8).permute(0, 3, 1, 2, 4).reshape(rb * 32, K)[:rows]
    assert torch.equal(first, x.to(torch.bfloat16).float())


@pytest.mark.parametrize("M,N,K", [(96, 256, 32), (100, 256, 256), (333, 96, 64), (2000, 512, 256),
                                   (21950, 1024, 256), (21950, 256, 1024), (4200, 544, 256), (333, 320, 64)])
def test_integer_problems_are_exact(hip, M, N, K):
    """Small-integer operands: every piece product and every partial sum is an integer below 2^24,
    so the result must equal the integer product exactly -- on every tile, ragged edges included,
    and identically in the fp32-row and the S3 output."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-8, 9, (M, K), generator=g).float().to(DEV)
    w = torch.randint(-8, 9, (N, K), generator=g).float().to(DEV)
    b = torch.randint(-50, 51, (N,), generator=g).float().to(DEV)
    want = (a.double() @ w.double().T + b.double()).float()
    out = torch.full((M, N), float("nan"), device=DEV)
    out_s = torch.empty(hip.s3_floats(M, N), device=DEV)
    hip.gemm_s3(s3(hip, a), s3(hip, w), M, N, K, bias=b, out=out, out_s3=out_s)
    assert torch.equal(out, want)
    assert torch.equal(joined(hip, out_s, M, N), want)
    # ReLU, and a wider row pitch (the [value | offsets | logits] rows)
    wide = torch.full((M, N + 32), -7.0, device=DEV)
    hip.gemm_s3(s3(hip, a), s3(hip, w), M, N, K, bias=b, relu=True, out=wide[:, :N])
    assert torch.equal(wide[:, :N], want.clamp_min(0))
    assert torch.all(wide[:, N:] == -7.0)
    if N >= 512:
        # N >= 512 takes the 192 x 256 tile (two row groups share W); the 96 x 256 tile on request
        out.fill_(float("nan"))
        hip.gemm_s3(s3(hip, a), s3(hip, w), M, N, K, bias=b, out=out, out_s3=out_s, tile96=True)
        assert torch.equal(out, want) and torch.equal(joined(hip, out_s, M, N), want)


@pytest.mark.parametrize("M,N,K,relu", [(21950, 1024, 256, True), (21950, 256, 1024, False),
                                        (21950, 512, 256, False), (1050, 256, 256, False)])
def test_error_against_fp64_is_at_or_below_the_fp32_mfma_kernel(hip, M, N, K, relu):
    """N(0,1) activations, N(0, 1/16) weights (the encoder's scales): max and rms error of the
    six-product form against an fp64 product, beside the exact-fp32 MFMA kernel's on the same
    operands.  Bar: rms <= 1.1 x the fp32 kernel's, max <= 2 x, and max <= 2e-5 of the output
    scale (the path's bar on logits is 1e-3)."""
    a, w, b = G(M, K, seed=3), G(N, K, seed=4, scale=1 / 16), G(N, seed=5)
    ref = a.double() @ w.double().T + b.double()
    if relu:
        ref = ref.clamp_min(0)
    got = torch.empty(M, N, device=DEV)
    hip.gemm_s3(s3(hip, a), s3(hip, w), M, N, K, bias=b, relu=relu, out=got)
    f32 = torch.empty(M, N, device=DEV)
    hip.linear(a, w, b, f32, relu=relu, force="tile64")     # the 64x64 fp32-MFMA tile kernel
    e_s3, e_32 = (got.double() - ref).abs(), (f32.double() - ref).abs()
    scale = max(1.0, ref.abs().max().item())
    print("M %d N %d K %d: s3 max %.3e rms %.3e | fp32 MFMA max %.3e rms %.3e" % (
        M, N, K, e_s3.max(), e_s3.pow(2).mean().sqrt(), e_32.max(), e_32.pow(2).mean().sqrt()))
    assert math.isfinite(e_s3.max().item())
    assert e_s3.pow(2).mean().sqrt() <= 1.1 * e_32.pow(2).mean().sqrt()
    assert e_s3.max() <= 2.0 * e_32.max() and e_s3.max() <= 2e-5 * scale


def test_second_operand_feeds_the_columns_from_a2_from_col_on(hip):
    """The encoder's [value | offsets | logits] projection: value columns read the tokens,
    the others tokens + positions (`query + query_pos`, facebook_detr.py:329-332)."""
    M, K, N = 4200, 256, 512
    x, pos, w, b = G(M, K, seed=6), G(1050, K, seed=7), G(N, K, seed=8, scale=1 / 16), G(N, seed=9)
    got = torch.empty(M, N, device=DEV)
    hip.gemm_s3(s3(hip, x), s3(hip, w), M, N, K, bias=b, out=got, a2=s3(hip, x, pos), a2_from_col=256)
    xp = x + pos[torch.arange(M, device=DEV) % 1050]
    ref = torch.cat([x.double() @ w[:256].double().T, xp.double() @ w[256:].double().T], 1) + b.double()
    assert (got.double() - ref).abs().max() <= 1e-5 * ref.abs().max()


@pytest.mark.parametrize("M,K", [(21950, 256), (21950, 1024), (100, 256), (4099, 512)])
def test_residual_layernorm_row_epilogue(hip, M, K):
    """out = LayerNorm(x W^T + b + res) gamma + beta with the residual read as an S3 operand (its
    exact fp32 values), against torch in fp64 (2e-5 of the output scale, the bar of the fp32 fused
    kernel's test), and against the fp32 kernel pair on the same operands (both are fp32 roundings
    of the same real number: 5e-6).  All three output forms carry the same bits; the `+ pos` form
    is split(out + pos[row % len])."""
    x, w, b = G(M, K, seed=10), G(256, K, seed=11, scale=0.2 / math.sqrt(K / 256)), G(256, seed=12)
    res, pos = G(M, 256, seed=13), G(max(1, M // 2), 256, seed=14)
    g, be = (G(256, seed=15) * 0.2 + 1.0), G(256, seed=16)
    out = torch.full((M, 256), float("nan"), device=DEV)
    out_s = torch.empty(hip.s3_floats(M, 256), device=DEV)
    out_p = torch.empty(hip.s3_floats(M, 256), device=DEV)
    hip.gemm_s3(s3(hip, x), s3(hip, w), M, 256, K, bias=b, out=out, out_s3=out_s, out_s3_pos=out_p,
                pos=(hip.pos8(pos), pos.shape[0]), res_s3=s3(hip, res), gamma=g, beta=be)
    ref = F.layer_norm(res.double() + x.double() @ w.double().T + b.double(), (256,), g.double(),
                       be.double(), 1e-5)
    err = (out.double() - ref).abs().max().item()
    assert math.isfinite(err) and err <= 2e-5 * max(1.0, ref.abs().max().item()), err
    pair = torch.empty(M, 256, device=DEV)
    hip.linear_res_ln(x, w, b, res, g, be, pair)
    assert (out - pair).abs().max() <= 5e-6 * max(1.0, ref.abs().max().item())
    assert torch.equal(joined(hip, out_s, M, 256), out)
    want_p = out + pos[torch.arange(M, device=DEV) % pos.shape[0]]
    assert torch.equal(joined(hip, out_p, M, 256), want_p)


def test_launches_are_deterministic_and_reject_bad_contracts(hip):
    M, N, K = 5000, 1024, 256
    a, w = s3(hip, G(M, K, seed=20)), s3(hip, G(N, K, seed=21))
    outs = []
    for _ in range(3):
        o = torch.empty(M, N, device=DEV)
        hip.gemm_s3(a, w, M, N, K, out=o)
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    o = torch.empty(M, N, device=DEV)
    with pytest.raises(RuntimeError):
        hip.gemm_s3(a, w, M, N, 48, out=o)            # K % 32
    with pytest.raises(RuntimeError):
        hip.gemm_s3(a, w, M, 100, K, out=o)           # N % 32
    with pytest.raises(RuntimeError):
        hip.gemm_s3(a, w, M, N, K)                    # no output
    with pytest.raises(RuntimeError):
        hip.gemm_s3(a, w, M, N, K, out=o, gamma=o, beta=o)   # LayerNorm needs N == 256


def test_sampling_kernel_writes_the_same_values_pre_split(hip):
    """pn_msda_ex_f32 with PN_MSDA_S3_OUT: output_proj's A operand straight from the sampling
    kernel -- bit for bit the split of its fp32 output (800x1333 pyramid and a small ragged one)."""
    for shapes in ([(25, 42), (50, 84), (100, 167)], [(3, 5), (6, 9), (12, 17)]):
        SN = sum(h * w for h, w in shapes)
        voa = G(1, SN, 544, seed=30)
        voa[..., 256:448] *= 3.0
        ref = torch.full((1, SN, 256), float("nan"), device=DEV)
        hip.msda(voa, 544, voa.view(-1)[256:], 544, ref, 1, shapes)
        out = torch.zeros(hip.s3_floats(SN, 256), device=DEV)
        hip.msda(voa, 544, voa.view(-1)[256:], 544, out, 1, shapes, s3_out=True)
        assert torch.equal(joined(hip, out, SN, 256), ref.view(SN, 256))


@pytest.mark.parametrize("rows,C", [(4200, 768), (100, 192), (33, 96), (1050, 1536), (70, 3072)])
def test_layernorm_rows_written_pre_split(hip, rows, C):
    """pn_layernorm_rows_s3_f32 (the Swin blocks' norm1 / norm2 in front of the bf16x3 GEMMs):
    bit for bit the split of pn_layernorm_rows_f32's fp32 rows."""
    x, g, b = G(rows, C, seed=40), G(C, seed=41) * 0.2 + 1.0, G(C, seed=42)
    ref = torch.empty(rows, C, device=DEV)
    hip.layernorm_rows(x, g, b, ref)
    out = torch.zeros(hip.s3_floats(rows, C), device=DEV)
    hip.layernorm_rows_s3(x, g, b, out)
    assert torch.equal(joined(hip, out, rows, C), ref)


@pytest.mark.parametrize("M,N,K", [(4200, 768, 768), (4200, 3072, 768), (1050, 1536, 6144), (333, 192, 192),
                                   (4200, 2304, 768)])
def test_gelu_and_fp32_shortcut_epilogues(hip, M, N, K):
    """The Swin blocks' forms: out = gelu(x W^T + b) (exact erf GELU, S3 output) and
    out = x W^T + b + res with the shortcut read from fp32 rows, in place (`out is res`).
    Against torch in fp64: 2e-5 of the output scale."""
    x, w, b = G(M, K, seed=50), G(N, K, seed=51, scale=0.5 / math.sqrt(K)), G(N, seed=52)
    x_s, w_s = s3(hip, x), s3(hip, w)
    out_s = torch.empty(hip.s3_floats(M, N), device=DEV)
    hip.gemm_s3(x_s, w_s, M, N, K, bias=b, gelu=True, out_s3=out_s)
    ref = F.gelu(x.double() @ w.double().T + b.double())
    got = joined(hip, out_s, M, N)
    assert (got.double() - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max().item())
    res = G(M, N, seed=53)
    ref = x.double() @ w.double().T + b.double() + res.double()
    inplace = res.clone()
    hip.gemm_s3(x_s, w_s, M, N, K, bias=b, out=inplace, res=inplace)
    assert (inplace.double() - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max().item())
    for kw in (dict(tile96=True), dict(tile192=True), dict()):
        sep = torch.empty(M, N, device=DEV)
        hip.gemm_s3(x_s, w_s, M, N, K, bias=b, out=sep, res=res, **kw)
        assert torch.equal(sep, inplace)
