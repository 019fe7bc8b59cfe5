"""GPU: the backward of Pair-Net's own tail (pair-net_amd/grad.py + csrc/grad.hip; SURVEY 8 f-4,
second slice) against torch autograd through the reference-pinned oracle (oracle/head.py is the
arithmetic of pairnet_head.py bit for bit, tests/test_oracle.py).  The oracle runs in float64
for the gradient reference: both fp32 implementations then sit a rounding error away from it.
Tolerance: 1e-4 of the largest entry of each gradient tensor (`query`, pair features and every
parameter), as VERDICT r5 next 6 asks; the taped forward must reproduce the recorded / inference
outputs to 1e-4 absolute."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import golden, head_cfg, oracle_head, overrides_of
from oracle.head import OracleCrossHead2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def _hip_head(sd):
    from pairnet_amd import CrossHead2
    head = CrossHead2(**head_cfg())
    head.load_state_dict(sd)
    return head.to(DEV)


def _oracle64(sd):
    head = OracleCrossHead2(**head_cfg()).eval()
    head.load_state_dict({k: v.detach().cpu() for k, v in sd.items()}, strict=True)
    return head.double()


def _tail(head_o, q, sub_pos, obj_pos, cls_detached=False):
    """oracle/head.py forward's tail (pairnet_head.py:322-392) with autograd ON: q [Q, B, 256].
    `cls_detached`: gather sub / obj from `cls_pred.clone().detach()` as the reference does."""
    x = head_o.transformer_decoder.post_norm(q).transpose(0, 1)
    cls = head_o.cls_embed(x)
    if cls_detached:
        cls = cls.clone().detach()
    s = F.normalize(head_o.sub_query_update(q).transpose(0, 1), p=2, dim=-1, eps=1e-12)
    o = F.normalize(head_o.obj_query_update(q).transpose(0, 1), p=2, dim=-1, eps=1e-12)
    raw = torch.matmul(s, o.transpose(1, 2))
    imp = head_o.update_importance(raw)
    pair, rel = OracleCrossHead2.relation_logits.__wrapped__(head_o, q, sub_pos, obj_pos)
    nc = cls.shape[-1]
    g = lambda p: torch.gather(cls, 1, p.unsqueeze(-1).expand(-1, -1, nc))
    return dict(rel=rel, importance=imp, importance_raw=raw, sub=g(sub_pos), obj=g(obj_pos),
                cls=cls)


def _compare(name, got, ref, report):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert tuple(got.shape) == tuple(ref.shape), (name, got.shape, ref.shape)
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    report.append((name, scale, err))
    if scale == 0.0:
        assert err == 0.0, (name, err)
    else:
        assert err <= TOL * scale, (name, err, scale)


def _compare_params(grads, head_o, report, expect):
    ref = {k: p.grad for k, p in head_o.named_parameters() if p.grad is not None}
    for k in expect:
        assert k in grads and k in ref, k
        _compare(k, grads[k], ref[k], report)
    # nothing the oracle differentiated is missing from the GPU's dict
    missing = [k for k, g in ref.items() if float(g.abs().max()) > 0 and k not in grads]
    assert not missing, missing


def _print(report):
    worst = max(report, key=lambda r: r[2] / r[1] if r[1] else 0.0)
    print("%d gradient tensors; worst relative error %.2e (%s, max |g| %.3e)"
          % (len(report), worst[2] / worst[1], worst[0], worst[1]))


def _rel_names(L=6):
    names = ["rel_cls_embed.weight", "rel_cls_embed.bias", "rel_query_feat.weight",
             "rel_query_embed.weight", "rel_query_embed2.weight"]
    for i in range(L):
        pre = "relation_decoder.layers.%d." % i
        for a in ("attentions.0.attn.", "attentions.1.attn."):
            names += [pre + a + n for n in ("in_proj_weight", "in_proj_bias", "out_proj.weight",
                                            "out_proj.bias")]
        names += [pre + "norms.%d.%s" % (j, n) for j in range(3) for n in ("weight", "bias")]
        names += [pre + "ffns.0.layers.0.0.weight", pre + "ffns.0.layers.0.0.bias",
                  pre + "ffns.0.layers.1.weight", pre + "ffns.0.layers.1.bias"]
    return names


def _ppn_names():
    names = ["%s_query_update.%d.%s" % (s, j, n) for s in ("sub", "obj") for j in (0, 2, 4)
             for n in ("weight", "bias")]
    names += ["update_importance.conv_layers.%d.0.%s" % (j, n) for j in range(3)
              for n in ("weight", "bias")]
    return names


def test_building_blocks_against_torch():
    """csrc/grad.hip's small kernels one by one against their torch expressions."""
    from pairnet_amd import hip
    g = torch.Generator().manual_seed(0)
    R = lambda *s: torch.randn(*s, generator=g).to(DEV)
    x = R(100, 56)
    xt = torch.empty(56, 104, device=DEV)
    hip.transpose(x, xt)
    assert torch.equal(xt[:, :100], x.t()) and float(xt[:, 100:].abs().max()) == 0.0
    big = R(5000, 200)
    cs = torch.full((200,), 3.0, device=DEV)
    hip.colsum(big, cs, accumulate=True)
    assert float((cs - (big.double().sum(0) + 3.0).float()).abs().max()) < 2e-3
    y, dy = R(777), R(777)
    dx = torch.empty_like(dy)
    hip.relu_bwd(dy, y, dx)
    assert torch.equal(dx, torch.where(y > 0, dy, torch.zeros_like(dy)))
    a, b = R(6, 100, 256), R(100, 256)
    out = torch.empty_like(a)
    hip.add_periodic(a, b, out)
    assert torch.equal(out, a + b)
    acc = R(100, 256)
    want = acc + (a[0] + a[1] + a[2] + a[3] + a[4] + a[5])
    hip.batch_sum(a, acc, 6, accumulate=True)
    assert float((acc - want).abs().max()) < 1e-5
    # LayerNorm backward
    xin = (R(300, 256) * 2 + 0.5).requires_grad_()
    gam, bet = R(256), R(256)
    up = R(300, 256)
    gl, bl = gam.clone().requires_grad_(), bet.clone().requires_grad_()
    F.layer_norm(xin, (256,), gl, bl, 1e-5).backward(up)
    dxk, gx = torch.empty(300, 256, device=DEV), torch.empty(300, 256, device=DEV)
    hip.layernorm256_bwd(up, xin.detach(), gam, dxk, gx)
    dg = torch.zeros(256, device=DEV)
    hip.colsum(gx, dg)
    assert float((dxk - xin.grad).abs().max()) < 1e-4 * float(xin.grad.abs().max())
    assert float((dg - gl.grad).abs().max()) < 1e-4 * float(gl.grad.abs().max())
    # attention backward (8 heads x 32) against torch's scaled_dot_product_attention
    B, Nq, Nk = 2, 100, 200
    q, k, v = (R(B * n, 256).requires_grad_() for n in (Nq, Nk, Nk))
    do = R(B * Nq, 256)
    hd = lambda t, n: t.view(B, n, 8, 32).transpose(1, 2)
    o = F.scaled_dot_product_attention(hd(q, Nq), hd(k, Nk), hd(v, Nk))
    o.transpose(1, 2).reshape(B * Nq, 256).backward(do)
    dq, dk, dv = (torch.empty_like(t) for t in (q, k, v))
    scr = torch.empty(hip.mha_bwd_scratch_floats(B, Nq, Nk), device=DEV)
    hip.mha_bwd(q.detach(), k.detach(), v.detach(), do, dq, dk, dv, scr, B, Nq, Nk, 32 ** -0.5)
    for got, ref in ((dq, q.grad), (dk, k.grad), (dv, v.grad)):
        assert float((got - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    # scatter = the transpose of gather
    idx = torch.randint(0, 100, (2, 200), generator=g).to(DEV)
    src = R(400, 256)
    outp = torch.zeros(200, 256, device=DEV)
    hip.scatter_rows_add(src, idx, outp, 2, 100, 200, 256)
    want = torch.zeros(2, 100, 256, device=DEV).index_put_(
        (torch.arange(2, device=DEV)[:, None].expand(2, 200), idx), src.view(2, 200, 256),
        accumulate=True)
    assert float((outp.view(2, 100, 256) - want).abs().max()) < 1e-5


def test_relation_decoder_backward_on_golden_pair_features():
    """`reldec.npz`'s pair features through the taped Relation Fusion decoder: the forward
    reproduces the reference's recorded rel_preds, the backward equals autograd through the
    oracle's six layers for a random upstream gradient (d pair features and all 113 parameter
    tensors of relation_decoder.* / rel_cls_embed / rel_query_*)."""
    from pairnet_amd import RelationTailGrad
    fx = golden("reldec")
    _, sd, _ = oracle_head(int(fx["weight_seed"]))
    head = _hip_head(sd)
    pair_seq = torch.from_numpy(fx["pair_feat"])                   # (2R, B, 256) seq-first
    B = pair_seq.shape[1]
    tape = RelationTailGrad(head)
    rel = tape.relation_forward(pair_seq.transpose(0, 1).reshape(-1, 256).to(DEV))
    torch.cuda.synchronize()
    assert float((rel.cpu() - torch.from_numpy(fx["rel_preds"])).abs().max()) < 1e-4
    g = torch.randn(rel.shape, generator=torch.Generator().manual_seed(11))
    dpair, grads = tape.relation_backward(g)
    torch.cuda.synchronize()

    head_o = _oracle64(sd)
    pair_o = pair_seq.double().requires_grad_()
    r = head_o.rel_query_feat.weight.unsqueeze(1).repeat((1, B, 1))
    r_pos = head_o.rel_query_embed.weight.unsqueeze(1).repeat((1, B, 1))
    p_pos = head_o.rel_query_embed2.weight.unsqueeze(1).repeat((1, B, 1))
    for layer in head_o.relation_decoder.layers:          # (oracle/head.py relation_logits)
        r = layer(query=r, key=pair_o, value=pair_o, query_pos=r_pos, key_pos=p_pos,
                  query_key_padding_mask=None, key_padding_mask=None)
    rel_o = head_o.rel_cls_embed(r.transpose(0, 1))
    (rel_o * g.double()).sum().backward()
    report = []
    _compare("pair_feat", dpair.view(B, -1, 256).transpose(0, 1), pair_o.grad, report)
    _compare_params(grads, head_o, report, _rel_names())
    _print(report)


def test_pair_proposal_backward_on_golden_queries():
    """`ppn_sep.npz`'s decoder queries through the taped PPN (two MLPs, F.normalize, cosine
    matrix, ConvTiny): recorded importance reproduced, top-k list exact, and d importance ->
    (d queries, the twelve MLP and six convolution parameter gradients) equal autograd."""
    from pairnet_amd import RelationTailGrad
    fx = golden("ppn_sep")
    _, sd, _ = oracle_head(int(fx["weight_seed"]), overrides_of(fx))
    head = _hip_head(sd)
    q_seq = torch.from_numpy(fx["query_feat"])                     # (Q, B, 256)
    Q, B = q_seq.shape[:2]
    tape = RelationTailGrad(head)
    out = tape.forward(q_seq.transpose(0, 1).reshape(-1, 256).contiguous().to(DEV))
    torch.cuda.synchronize()
    assert float((out["importance"].cpu() - torch.from_numpy(fx["importance"])).abs().max()) < 1e-3
    assert np.array_equal(out["sub_pos"].cpu().numpy(), fx["sub_pos"].reshape(B, -1))
    assert np.array_equal(out["obj_pos"].cpu().numpy(), fx["obj_pos"].reshape(B, -1))
    g = torch.randn(B, Q, Q, generator=torch.Generator().manual_seed(12))
    dq, grads = tape.backward(g_importance=g)
    torch.cuda.synchronize()

    head_o = _oracle64(sd)
    q_o = q_seq.double().requires_grad_()
    o = _tail(head_o, q_o, out["sub_pos"].cpu(), out["obj_pos"].cpu())
    (o["importance"] * g.double()).sum().backward()
    report = []
    _compare("query", dq.view(B, Q, 256).transpose(0, 1), q_o.grad, report)
    _compare_params(grads, head_o, report, _ppn_names())
    _print(report)


@pytest.mark.parametrize("cls_detached", [True, False])
def test_tail_backward_from_the_losses_at_800x1333(cls_detached):
    """The whole slice at the bench's size (an 800x1333 image padded to 800x1344): head forward,
    `CrossHead2.loss(grads=)` gives d loss / d {rel, importance, sub, obj}; the taped tail re-runs
    from the plan's decoder queries (same outputs as the inference kernels to 1e-4), and its
    backward -- d queries and every parameter between `post_norm` and `rel_cls_embed` -- equals
    autograd through the oracle's tail on the same queries and the same selected pairs.
    `cls_detached=True` is the reference's graph (pairnet_head.py:380-390: the class logits are
    detached before the subject / object gathers, so two of the four loss terms train nothing);
    False checks the un-detached derivative through the gathers, `cls_embed` and `post_norm`."""
    from pairnet_amd import RelationTailGrad
    from test_losses_gpu import _outputs
    head, cls, masks, metas, gt_rels, gt_labels, gt_masks, pts = _outputs(1, H=800, W=1344, bs=1)
    up = {}
    head.loss(cls, masks, gt_rels, None, gt_labels, gt_masks, metas, point_coords=pts, grads=up)
    pl = head._last_plan
    Q = head.num_obj_query
    q = pl.q.clone()
    B = q.shape[0] // Q
    tape = RelationTailGrad(head)
    out = tape.forward(q, pl.sub_pos, pl.obj_pos)
    torch.cuda.synchronize()
    for k in ("rel", "importance", "sub", "obj", "cls"):
        assert float((out[k] - cls[k]).abs().max()) < 1e-4, k
    dq, grads = tape.backward(g_rel=up["rel"], g_importance=up["importance"], g_sub=up["sub"],
                              g_obj=up["obj"], cls_detached=cls_detached)
    torch.cuda.synchronize()

    head_o = _oracle64(head.state_dict())
    q_o = q.cpu().double().view(B, Q, 256).transpose(0, 1).contiguous().requires_grad_()
    o = _tail(head_o, q_o, pl.sub_pos.cpu(), pl.obj_pos.cpu(), cls_detached)
    sum((o[k] * up[k].cpu().double()).sum() for k in ("rel", "importance", "sub", "obj")).backward()
    report = []
    _compare("query", dq.view(B, Q, 256).transpose(0, 1), q_o.grad, report)
    names = _rel_names() + _ppn_names()
    cls_names = ["cls_embed.weight", "cls_embed.bias", "transformer_decoder.post_norm.weight",
                 "transformer_decoder.post_norm.bias"]
    if cls_detached:
        for k in cls_names:
            assert float(grads[k].abs().max()) == 0.0, k
    else:
        names += cls_names
    _compare_params(grads, head_o, report, names)
    _print(report)


def _unpack_mask(bits, rowall, B, Q, N):
    """The boolean attention mask (True = not attendable) of one layer from pn_mask_pack's bits,
    with the reference's all-masked fix (pairnet_head.py:300) applied."""
    nw = (N + 31) // 32
    words = bits.view(B * Q, nw).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    j = np.arange(N)
    m = ((words[:, j >> 5] >> (j & 31)) & 1).astype(bool)
    m[rowall.cpu().numpy().astype(bool)] = False
    return torch.from_numpy(m).view(B, Q, N)


@pytest.mark.parametrize("exact_mask_order", [True, False, "full"])
def test_masked_decoder_backward_against_the_oracle(exact_mask_order):
    """`HeadGrad`: the nine masked-attention decoder layers + the tail.  The taped query chain
    reproduces the inference kernels' outputs; its backward -- d memory tokens (what the pixel
    decoder's backward would receive), `query_feat`, `query_embed`, `level_embed`, the 9 x 18
    decoder tensors and the tail -- equals autograd through the oracle's layers fed with the SAME
    memory tokens and the SAME boolean masks (the masks are `detach()`ed thresholds,
    pairnet_head.py:256: a logit within rounding of 0 may flip between two fp32 implementations,
    which is a different function, not a gradient error).  A level of 6 keys (2 x 3) exercises
    the padded contraction of the K / V projection gradients."""
    from oracle import seeded
    from pairnet_amd import HeadGrad
    _, sd, _ = oracle_head(1234)
    head = _hip_head(sd)
    head.exact_mask_order = exact_mask_order       # (the three forms of the attention-mask step)
    H, W = 64, 96
    feats = seeded.seeded_feats(99, 2, H, W)
    metas = [dict(img_shape=(H, W, 3), scale_factor=[1.5] * 4)] * 2
    cls, _ = head.forward([f.to(DEV) for f in feats], metas)
    pl = head._last_plan
    B, Q = pl.B, head.num_obj_query
    ref_out = {k: cls[k].clone() for k in ("rel", "importance")}
    tape = HeadGrad(head)
    out = tape.forward_from_plan(pl, pl.sub_pos, pl.obj_pos)
    torch.cuda.synchronize()
    for k in ("rel", "importance"):
        assert float((out[k] - ref_out[k]).abs().max()) < 1e-4, k
    gen = torch.Generator().manual_seed(13)
    g1, g2 = torch.randn(out["rel"].shape, generator=gen), torch.randn(B, Q, Q, generator=gen)
    dmem, grads = tape.backward(g_rel=g1, g_importance=g2)
    torch.cuda.synchronize()

    head_o = _oracle64(sd)
    mem = pl.X.cpu().double().requires_grad_()                     # [B, SN, 256]
    keys, key_pos = [], []
    for l in range(3):
        h, w = pl.shapes[l]
        m = mem[:, pl.start[l]:pl.start[l] + pl.N[l]].transpose(0, 1)
        keys.append(m + head_o.level_embed.weight[l].view(1, 1, -1))
        pad = torch.zeros((B, h, w), dtype=torch.bool)
        key_pos.append(head_o.decoder_positional_encoding(pad).flatten(2).permute(2, 0, 1).double())
    q = head_o.query_feat.weight.unsqueeze(1).repeat((1, B, 1))
    q_pos = head_o.query_embed.weight.unsqueeze(1).repeat((1, B, 1))
    for i, layer in enumerate(head_o.transformer_decoder.layers):
        s = tape.dt["layers"][i]
        l = i % 3
        mask = _unpack_mask(s["bits"], s["rowall"], B, Q, pl.N[l])
        mask = mask.unsqueeze(1).repeat((1, head_o.n_heads, 1, 1)).flatten(0, 1)
        q = layer(query=q, key=keys[l], value=keys[l], query_pos=q_pos, key_pos=key_pos[l],
                  attn_masks=[mask, None], query_key_padding_mask=None, key_padding_mask=None)
    o = _tail(head_o, q, pl.sub_pos.cpu(), pl.obj_pos.cpu())
    ((o["rel"] * g1.double()).sum() + (o["importance"] * g2.double()).sum()).backward()
    report = []
    _compare("memory tokens", dmem, mem.grad, report)
    names = _rel_names() + _ppn_names() + ["query_feat.weight", "query_embed.weight",
                                           "level_embed.weight"]
    from pairnet_amd.grad import RelationTailGrad
    for i in range(9):
        names += RelationTailGrad._layer_names("transformer_decoder.layers.%d." % i)
    _compare_params(grads, head_o, report, names)
    _print(report)


def test_pixel_decoder_backward_against_the_oracle():
    """`PixelDecoderGrad`: backbone features -> 1x1 input convolutions + GroupNorm -> six
    deformable-attention encoder layers -> memory tokens, taped on the exact-fp32 kernels; the
    backward (sampling operator in mmcv's operand set, its operands' projections, LayerNorm / FFN,
    GroupNorm, the convolutions) against autograd through the HF-pinned oracle pixel decoder in
    float64: d features (C3, C4, C5) and all 109 parameter tensors."""
    from oracle import seeded
    from pairnet_amd import PixelDecoderGrad
    _, sd, _ = oracle_head(1234)
    head = _hip_head(sd)
    H, W = 64, 96
    feats = seeded.seeded_feats(99, 2, H, W)
    tape = PixelDecoderGrad(head)
    mem = tape.forward([f.to(DEV) for f in feats])
    torch.cuda.synchronize()
    head_o = _oracle64(sd)
    f64 = [f.double().requires_grad_() for f in feats]
    _, memories = head_o.pixel_decoder(f64)
    mem_ref = torch.cat([m.flatten(2).transpose(1, 2) for m in memories], 1)      # [B, SN, 256]
    assert float((mem.cpu().double() - mem_ref.detach()).abs().max()) < 1e-4
    G = torch.randn(mem_ref.shape, generator=torch.Generator().manual_seed(14))
    dfeats, grads = tape.backward(G)
    torch.cuda.synchronize()
    (mem_ref * G.double()).sum().backward()
    report = []
    for l in range(3):
        _compare("features of level %d" % l, dfeats[l], f64[3 - l].grad, report)
    names = [n for _, ns in PixelDecoderGrad.param_groups(head) for n in ns]
    _compare_params(grads, head_o, report, names)
    assert f64[0].grad is None or float(f64[0].grad.abs().max()) == 0.0     # C2 feeds the mask branch only
    _print(report)


def test_backbone_backward_against_the_oracle():
    """`BackboneGrad`: ResNet-50 stages 2-4 with frozen (folded) BatchNorm from C2 to C3 / C4 / C5:
    the taped forward equals the inference backbone's outputs, the weight gradients of all 42
    trainable convolutions (1x1, 3x3 at stride 1 and 2, projection shortcuts) equal autograd
    through the oracle ResNet (float64, BatchNorm in eval mode) fed with the same C2."""
    from oracle.backbone import OracleResNet50, seeded_backbone_state
    from pairnet_amd import BackboneGrad, ResNet50Hip
    sd = seeded_backbone_state(5)
    bb = ResNet50Hip()
    bb.load_state_dict(sd)
    bb.to(DEV)
    img = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(15)).to(DEV)
    c2, c3, c4, c5 = [f.clone() for f in bb(img)]
    tape = BackboneGrad(bb)
    outs = tape.forward(c2)
    torch.cuda.synchronize()
    for got, ref in zip(outs, (c3, c4, c5)):
        scale = float(ref.abs().max())
        assert float((got.permute(0, 3, 1, 2) - ref).abs().max()) < 1e-4 * max(1.0, scale)
    gen = torch.Generator().manual_seed(16)
    G = [torch.randn(tuple(f.shape), generator=gen) for f in (c3, c4, c5)]
    grads = tape.backward(*G)
    torch.cuda.synchronize()

    o = OracleResNet50()
    o.load_state_dict(sd)
    o = o.double().eval()
    x = c2.cpu().double().contiguous()
    loss = 0.0
    for i, g in zip((2, 3, 4), G):
        x = getattr(o, "layer%d" % i)(x)
        loss = loss + (x * g.double()).sum()
    loss.backward()
    ref = dict(o.named_parameters())
    report = []
    names = [n for _, ns in BackboneGrad.param_groups(bb) for n in ns]
    assert len(names) == 42
    for n in names:
        _compare(n, grads[n], ref[n].grad, report)
    assert ref["layer1.0.conv1.weight"].grad is None          # (frozen stage: not even reached)
    _print(report)
