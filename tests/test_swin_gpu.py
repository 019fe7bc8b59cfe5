"""GPU: the native Swin backbone (BASELINE.json configs[3]) against
oracle/swin.py (torch CPU fp32 restatement of mmdet's SwinTransformer, itself pinned to
transformers.SwinBackbone by tests/test_oracle.py), and the kernels only it uses.

Tolerance: fp32; 2e-5 relative per kernel, 2e-4 of the feature scale for whole backbones
(up to 24 blocks of re-associated fp32 sums)."""
import pytest
import torch
import torch.nn.functional as F

from oracle.swin import (OracleSwin, PatchMerging, ShiftWindowMSA, seeded_swin_state)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def R(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def rel(a, b):
    b = b.double()
    return float((a.detach().cpu().double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("rows,C", [(5, 96), (1000, 128), (333, 192), (70, 384), (9, 1536),
                                    (130, 3072), (64, 256)])
def test_layernorm_rows_matches_torch(rows, C):
    from pairnet_amd import hip
    x, g, b = R(rows, C, seed=1, scale=3.0) + 0.7, R(C, seed=2), R(C, seed=3)
    want = F.layer_norm(x, (C,), g, b, 1e-5)
    out = torch.empty(rows, C, device=DEV)
    hip.layernorm_rows(x.to(DEV), g.to(DEV), b.to(DEV), out)
    assert rel(out, want) < 2e-6
    # strided views (a column block of a wider matrix)
    wide = torch.zeros(rows, C + 64, device=DEV)
    wide[:, :C] = x.to(DEV)
    out2 = torch.full((rows, C + 32), 7.0, device=DEV)
    hip.layernorm_rows(wide[:, :C], g.to(DEV), b.to(DEV), out2[:, :C])
    assert torch.equal(out2[:, :C], out) and bool((out2[:, C:] == 7.0).all())


def test_layernorm_rows_rejects_bad_widths():
    from pairnet_amd import hip
    x = torch.zeros(4, 3076, device=DEV)
    with pytest.raises(RuntimeError):
        hip.layernorm_rows(x, x[0], x[0], torch.empty_like(x))
    y = torch.zeros(4, 98, device=DEV)
    with pytest.raises(RuntimeError):
        hip.layernorm_rows(y, y[0], y[0], torch.empty_like(y))


@pytest.mark.parametrize("B,H,W,C", [(2, 6, 8, 32), (1, 7, 9, 96), (2, 25, 42, 64), (1, 5, 5, 768)])
def test_patch_merge_ln_matches_oracle(B, H, W, C):
    """2x2 gather (odd maps zero-padded) + LayerNorm(4C) with the weights permuted from
    mmdet's nn.Unfold order to the kernel's neighbour-major order."""
    from pairnet_amd import hip
    pm = PatchMerging(C)
    pm.norm.weight.data, pm.norm.bias.data = R(4 * C, seed=4) + 1.0, R(4 * C, seed=5)
    pm.reduction.weight.data = R(2 * C, 4 * C, seed=6, scale=(4 * C) ** -0.5)
    x = R(B, H * W, C, seed=7)
    with torch.no_grad():
        want, (h2, w2) = pm(x, (H, W))
    perm = lambda v: v.reshape(*v.shape[:-1], C, 4).transpose(-1, -2).reshape(v.shape).contiguous()
    mg = torch.empty(B * h2 * w2, 4 * C, device=DEV)
    hip.patch_merge_ln(x.to(DEV), perm(pm.norm.weight.data).to(DEV), perm(pm.norm.bias.data).to(DEV),
                       mg, B, H, W, C)
    out = torch.empty(B * h2 * w2, 2 * C, device=DEV)
    hip.linear(mg, perm(pm.reduction.weight.data).to(DEV), None, out)
    assert rel(out.view(B, h2 * w2, 2 * C), want) < 2e-5


@pytest.mark.parametrize("B,H,W", [(1, 64, 96), (2, 75, 101), (1, 5, 7)])
def test_patch_embedding_matches_conv(B, H, W):
    from pairnet_amd import hip
    img, w, b = R(B, 3, H, W, seed=8), R(96, 3, 4, 4, seed=9, scale=0.1), R(96, seed=10)
    want = F.conv2d(F.pad(img, (0, (4 - W % 4) % 4, 0, (4 - H % 4) % 4)), w, b, stride=4)
    h4, w4 = want.shape[-2:]
    cols = torch.empty(B * h4 * w4, 64, device=DEV)
    hip.patch_im2col4(img.to(DEV), cols, B, H, W)
    wp = torch.zeros(96, 64)
    wp[:, :48] = w.reshape(96, 48)
    out = torch.empty(B * h4 * w4, 96, device=DEV)
    hip.linear(cols, wp.to(DEV), b.to(DEV), out)
    assert rel(out.view(B, h4, w4, 96).permute(0, 3, 1, 2), want) < 2e-5


@pytest.mark.parametrize("M,N,K", [(100, 256, 128), (4200, 512, 128), (21950, 384, 96)])
def test_gemm_gelu_epilogue(M, N, K):
    """exact (erf) GELU epilogue, with and without a residual, on every kernel family."""
    from pairnet_amd import hip
    x, w, b, r = R(M, K, seed=1), R(N, K, seed=2, scale=K ** -0.5), R(N, seed=3), R(M, N, seed=4)
    want = F.gelu(x @ w.t() + b)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    for force in (None, "skinny", "tile64", "tile"):
        out = torch.empty(M, N, device=DEV)
        hip.linear(xd, wd, bd, out, gelu=True, force=force)
        assert rel(out, want) < 2e-5, force
    out = r.to(DEV)
    hip.linear(xd, wd, bd, out, gelu=True, res=out)        # residual added in place
    assert rel(out, want + r) < 2e-5


@pytest.mark.parametrize("B,H,W,heads,ws,shift", [
    (1, 14, 14, 2, 7, 0), (2, 14, 21, 1, 7, 3), (1, 13, 17, 3, 7, 3), (1, 24, 36, 2, 12, 0),
    (2, 25, 42, 4, 12, 6), (1, 5, 9, 1, 12, 6), (1, 8, 8, 2, 4, 2), (1, 50, 84, 1, 12, 6)])
def test_window_attention_matches_oracle(B, H, W, heads, ws, shift):
    """k_window_attn on the qkv rows == pad -> roll -> partition -> attention with relative
    position bias and shift mask -> merge -> roll back -> crop of the restatement."""
    from pairnet_amd import hip
    C = heads * 32
    m = ShiftWindowMSA(C, heads, ws, shift)
    m.w_msa.relative_position_bias_table.data = R((2 * ws - 1) ** 2, heads, seed=11)
    m.w_msa.qkv.weight.data, m.w_msa.qkv.bias.data = R(3 * C, C, seed=12, scale=C ** -0.5), R(3 * C, seed=13)
    m.w_msa.proj.weight.data, m.w_msa.proj.bias.data = torch.eye(C), torch.zeros(C)
    x = R(B, H * W, C, seed=14)
    with torch.no_grad():
        want = m(x, (H, W))
        qkv = m.w_msa.qkv(x).reshape(B * H * W, 3 * C)
    out = torch.full((B * H * W, C), float("nan"), device=DEV)
    hip.window_attention(qkv.to(DEV), m.w_msa.qkv.bias.data.to(DEV),
                         m.w_msa.relative_position_bias_table.data.t().contiguous().to(DEV), out, B, H, W, C, heads,
                         ws, shift)
    assert rel(out.view(B, H * W, C), want) < 2e-5
    # the same launch writing the proj GEMM's operand pre-split (pn_window_attention_s3_f32):
    # bit for bit the split of the fp32 rows
    n = B * H * W
    s3 = torch.zeros(hip.s3_floats(n, C), device=DEV)
    hip.window_attention_s3(qkv.to(DEV), m.w_msa.qkv.bias.data.to(DEV),
                            m.w_msa.relative_position_bias_table.data.t().contiguous().to(DEV), s3, B, H, W,
                            C, heads, ws, shift)
    back = torch.full((n, C), float("nan"), device=DEV)
    hip.s3_join(s3, back)
    assert torch.equal(back, out)


CONFIGS = {
    "tiny4": dict(embed_dims=32, depths=(2, 2, 2, 2), num_heads=(1, 2, 4, 8), window_size=4),
    "tiny7": dict(embed_dims=64, depths=(2, 2, 6, 2), num_heads=(2, 4, 8, 16), window_size=7),
    "b12": dict(embed_dims=128, depths=(2, 2, 4, 2), num_heads=(4, 8, 16, 32), window_size=12),
    # the TRUE models: the reference's Swin-B (configs/mask2former/pairnet_swinb.py:203-210)
    # and the Swin-L of BASELINE.json configs[3], 18 blocks in the third stage
    "swinB": dict(embed_dims=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=12),
    "swinL": dict(embed_dims=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), window_size=12),
}


@pytest.mark.parametrize("name,B,H,W", [("tiny4", 2, 75, 101), ("tiny7", 1, 128, 160),
                                        ("b12", 1, 200, 264), ("b12", 2, 96, 136),
                                        ("swinB", 1, 192, 256), ("swinL", 1, 192, 256),
                                        ("swinL", 2, 200, 264), ("swinB", 1, 800, 1333),
                                        ("swinL", 1, 800, 1333)])
def test_swin_backbone_matches_oracle(name, B, H, W):
    """(800 x 1333 with the reference's Swin-B: every stage map -- 200x334, 100x167, 50x84,
    25x42 -- needs window padding to a multiple of 12, and 167 / 25 are odd sides for the patch
    merging.)"""
    from pairnet_amd import SwinTransformerHip
    cfg = CONFIGS[name]
    oracle = OracleSwin(**cfg)
    sd = seeded_swin_state(oracle, 41)
    oracle.load_state_dict(sd)
    net = SwinTransformerHip(**cfg)
    assert set(net.state_dict()) == set(sd)
    net.load_state_dict(sd)
    net.to(DEV)
    img = R(B, 3, H, W, seed=15)
    with torch.no_grad():
        want = oracle(img)
    got = net(img.to(DEV))
    torch.cuda.synchronize()
    assert len(got) == len(want) == 4
    for i, (g, o) in enumerate(zip(got, want)):
        assert tuple(g.shape) == tuple(o.shape)
        assert g.is_contiguous(memory_format=torch.channels_last)
        e = rel(g, o)
        print("stage %d %s rel err %.2e" % (i, tuple(o.shape), e))
        assert e < 2e-4


def test_swin_state_dict_checks():
    from pairnet_amd import SwinTransformerHip
    net = SwinTransformerHip(**CONFIGS["tiny4"])
    sd = net.state_dict()
    bad = dict(sd)
    bad["stages.0.blocks.0.attn.w_msa.relative_position_index"] = \
        sd["stages.0.blocks.0.attn.w_msa.relative_position_index"].flip(0)
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad)
    del bad["stages.0.blocks.0.attn.w_msa.relative_position_index"]
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad)
    with pytest.raises(RuntimeError):
        SwinTransformerHip(**CONFIGS["tiny4"]).to("cpu")(torch.zeros(1, 3, 8, 8))
    with pytest.raises(NotImplementedError):
        SwinTransformerHip(embed_dims=48, depths=(2,), num_heads=(1,))


def test_detector_end_to_end_swin_backbone():
    """PSGTr(SwinTransformer, CrossHead2) as pairnet_swinb.py configures it (reduced depth):
    image -> triplets, and the head consumes the backbone's channels_last maps."""
    from pairnet_amd import build_detector, pairnet_swin
    cfg = pairnet_swin("B")
    cfg["backbone"]["depths"] = [2, 2, 2, 2]
    det = build_detector(cfg).to(DEV)
    img = R(1, 3, 128, 160, seed=9).to(DEV)
    metas = [dict(img_shape=(128, 160, 3), scale_factor=[1.0] * 4)]
    feats = det.extract_feat(img)
    assert [f.shape[1] for f in feats] == [128, 256, 512, 1024]
    assert [tuple(f.shape[2:]) for f in feats] == [(32, 40), (16, 20), (8, 10), (4, 5)]
    res = det.simple_test(img, metas)
    assert len(res) == 1 and res[0].rel_dists.shape == (100, 57)
    assert res[0].masks.shape == (200, 128, 160)


def test_swin_detector_pipelined_over_mixed_shapes_equals_the_synchronous_path():
    """Round 5: the Swin backbone's plans are views of one arena per slot too, and it takes a
    slot per stage-A stream in the detector's pipeline.  Images of several shapes through
    `dist.multi_gpu_test` (two backbones side by side, query chains beside them) give the
    records of the synchronous `simple_test` path, bit for bit; reserved arenas never grow."""
    from pairnet_amd import build_detector, pairnet_swin
    from pairnet_amd.dist import multi_gpu_test, pack_triplets
    cfg = pairnet_swin("B")
    cfg["backbone"]["depths"] = [2, 2, 2, 2]
    det = build_detector(cfg).to(DEV)
    shapes = [(128, 160), (96, 128), (160, 128), (128, 160), (112, 144), (96, 128), (160, 128)]
    data = []
    for i, (H, W) in enumerate(shapes):
        data.append((R(1, 3, H, W, seed=40 + i).to(DEV),
                     [dict(img_shape=(H, W, 3), scale_factor=[1.0] * 4)]))
    got = det.reserve([(160, 160)], depth=3, orig_sizes=[(160, 160)])
    assert got == det.backbone.arena_bytes() + det.bbox_head.arena_bytes() > 0
    head = det.bbox_head
    want = []
    for img, metas in data:
        res = head.simple_test(det.extract_feat(img), metas)[0]
        sub, obj = head.pair_positions()
        want.append(pack_triplets(res[1].cpu(), res[7].cpu(), sub[0].cpu(), obj[0].cpu()))
    out = multi_gpu_test(det, data, depth=3, calibrate=False)
    for i, w in enumerate(want):
        assert torch.equal(out["records"][i].cpu(), w), i
    assert sorted(det.backbone._arenas) == [0, 1]            # a slot per stage-A stream
    assert all(a.grows == 1 for a in det.backbone._arenas.values())
    assert all(a.grows == 1 for a in head._arenas.values())
