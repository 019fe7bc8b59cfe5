"""GPU: the whole hot path (CrossHead2.forward / get_bboxes through the C ABI) against
the golden vectors recorded from the reference and against the CPU oracle.

Tolerances (BASELINE.json north_star): relation logits within 1e-3 (fp32); top-k pair
indices bit-exact.  Strict index equality is asserted end to end on the SEPARATED
fixtures (e2e_small_sep at 96x128, e2e_full_sep at 800x1333, ppn_sep): seeded weights plus
stored edits that spread the reference's top scores far wider than fp32 rounding can move
them (oracle/make_golden.py `separate`; the generator asserts min_gap >= 1e-4 and >= 10 x
the reference's own fp32-vs-fp64 score difference).  On the older purely-random-weight
fixtures, whose smallest top-k gap (3e-8 ... 2e-7) is BELOW the reference's own
fp32-vs-fp64 difference, the list is compared tie-aware with TIE_TOL = 10 x the measured
score error and a floor on the number of identical positions."""
import numpy as np
import pytest
import torch

from helpers import golden, head_cfg, oracle_head, overrides_of, tie_aware_topk_match
from oracle import seeded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TIE_TOL_SMALL, TIE_TOL_FULL = 3e-6, 1.5e-5   # 10 x the GPU-vs-reference score error measured
                                              # on these fixtures (2.7e-7 / 1.5e-6)


def _hip_head(sd):
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from pairnet_amd import CrossHead2
    head = CrossHead2(**head_cfg())
    head.load_state_dict(sd)
    return head.to(DEV)


def _err(a, b):
    b = b.detach() if isinstance(b, torch.Tensor) else b
    return float((a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


def _rel_err(head_o, head, cls, ref_rel, ref_idx, trace_q):
    """Relation logits vs the reference.  The relation decoder is position-sensitive, so
    when a near-tie reorders the GPU's top-k list the reference logits are re-evaluated
    (CPU oracle, reference arithmetic) for the pair list the GPU selected."""
    pl = head._last_plan
    got_idx = pl.topk_idx.cpu()
    if torch.equal(got_idx, torch.as_tensor(ref_idx)):
        return _err(cls["rel"], ref_rel), True
    _, rel = head_o.relation_logits(trace_q, pl.sub_pos.cpu(), pl.obj_pos.cpu())
    return _err(cls["rel"], rel), False


def test_e2e_small_against_reference_golden():
    fx = golden("e2e_small")
    head_o, sd, crc = oracle_head(int(fx["weight_seed"]), overrides_of(fx))
    assert crc == int(fx["weight_crc"])
    H, W, bs = int(fx["height"]), int(fx["width"]), int(fx["batch"])
    feats = seeded.seeded_feats(int(fx["feat_seed"]), bs, H, W)
    assert seeded.checksum(feats) == int(fx["feat_crc"])
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0, 2.0, 2.0, 2.0])] * bs
    head = _hip_head(sd)
    cls, masks = head.forward([f.to(DEV) for f in feats], metas)
    torch.cuda.synchronize()
    trace = {}
    head_o.forward(feats, metas, trace=trace)
    errs = {k: _err(cls[k], fx["cls_" + k]) for k in ("cls", "importance")}
    errs["rel"], same = _rel_err(head_o, head, cls, fx["cls_rel"], fx["topk_idx"],
                                 trace["query_feat"])
    errs["mask"] = _err(masks["mask"], fx["mask_mask"])
    print("e2e_small errors:", errs, "top-k identical:", same)
    assert errs["rel"] < 1e-3 and errs["cls"] < 1e-3 and errs["importance"] < 1e-3
    assert errs["mask"] < 1e-3 * max(1.0, float(np.abs(fx["mask_mask"]).max()))
    pl = head._last_plan
    for b in range(bs):
        ok, exact = tie_aware_topk_match(fx["cls_importance"][b], fx["topk_idx"][b],
                                         pl.topk_idx[b].cpu().numpy(), TIE_TOL_SMALL)
        print("image %d: %d/100 top-k positions identical" % (b, exact))
        assert ok and exact >= 98
    # gathered outputs are consistent with the GPU's own indices
    sub = pl.sub_pos.cpu()
    assert torch.equal(cls["sub"].cpu(),
                       torch.gather(cls["cls"].cpu(), 1, sub[..., None].expand(-1, -1, 134)))
    # post-processing on the device vs the reference's tuple
    res = head.get_bboxes(cls, masks, metas)
    for i, r in enumerate(res):
        same_pairs = pl.topk_idx[i].cpu().numpy() == fx["topk_idx"][i]
        both = np.concatenate([same_pairs, same_pairs])
        assert np.array_equal(r[1].cpu().numpy()[both], fx["res%d_labels" % i][both])
        assert _err(torch.from_numpy(r[7].cpu().numpy()[same_pairs]),
                    fx["res%d_r_dists" % i][same_pairs]) < 1e-3
        shape = tuple(fx["res%d_masks_shape" % i])
        ref_masks = np.unpackbits(fx["res%d_masks" % i])[:int(np.prod(shape))].reshape(shape)
        got = r[3].cpu().numpy()
        assert got.shape == shape and got.dtype == np.bool_
        mism = (got[both] != ref_masks[both].astype(bool)).mean()
        print("image %d: mask bit mismatch %.2e, pan mismatch %.2e" % (
            i, mism, (r[4].cpu().numpy() != fx["res%d_pan_img" % i]).mean()))
        assert mism < 1e-3
        assert (r[4].cpu().numpy() != fx["res%d_pan_img" % i]).mean() < 5e-3
        assert np.array_equal(r[2].numpy(), fx["res%d_rel_pairs" % i])
        assert r[0].shape == (200, 5) and float(r[0].abs().sum()) == 0.0


@pytest.mark.parametrize("conv_algo,exact_order", [
    ("winograd", False), ("direct", False), ("winograd4", False), ("winograd", True),
    ("winograd4", True), ("winograd4", "full")])
def test_e2e_full_800x1333_against_reference_golden(conv_algo, exact_order):
    _e2e_full(conv_algo, exact_order)


def test_sparse_reference_mask_order_is_the_dense_one_bit_for_bit():
    """`exact_mask_order=True` computes, per layer, only the full-resolution mask logits the
    level's bilinear stencils read (Q x 4 N_l) and blends them; "full" computes all Q x H2 W2
    and resizes.  Same GEMM kernel per logit, same blend: every output of the head is bitwise
    the same, at 800 x 1333 with two images."""
    fx = golden("e2e_full")
    _, sd, _ = oracle_head(int(fx["weight_seed"]), overrides_of(fx))
    H, W = int(fx["height"]), int(fx["width"])
    feats = [f.to(DEV) for f in seeded.seeded_feats(int(fx["feat_seed"]) + 1, 2, H, W)]
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.083] * 4)] * 2
    outs = {}
    for mode in (True, "full", False):
        head = _hip_head(sd)
        head.exact_mask_order = mode
        cls, masks = head.forward(feats, metas)
        torch.cuda.synchronize()
        outs[mode] = {k: v.clone() for k, v in list(cls.items()) + list(masks.items())
                      if isinstance(v, torch.Tensor)}
        outs[mode]["topk"] = head._last_plan.topk_idx.clone()
    for k, v in outs[True].items():
        assert torch.equal(v, outs["full"][k]), k
    # (the once-resampled form is a different rounding of the same logits: it flips an
    # attention-mask bit only where a resized logit is within ~1e-6 of zero)
    print("outputs identical to the once-resampled form as well:",
          all(torch.equal(v, outs[False][k]) for k, v in outs[True].items()))


def _e2e_full(conv_algo, exact_order):
    fx = golden("e2e_full")
    head_o, sd, crc = oracle_head(int(fx["weight_seed"]), overrides_of(fx))
    assert crc == int(fx["weight_crc"])
    H, W = int(fx["height"]), int(fx["width"])
    feats = seeded.seeded_feats(int(fx["feat_seed"]), 1, H, W)
    assert seeded.checksum(feats) == int(fx["feat_crc"])
    assert [tuple(f.shape) for f in feats] == [tuple(s) for s in fx["feat_shapes"].tolist()]
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.083] * 4)]
    head = _hip_head(sd)
    head.conv_algo = conv_algo
    # (True / "full": attention masks in the reference's operation order -- full-size mask
    # logits, then the bilinear resize; sparse / dense evaluation -- instead of the
    # once-resampled mask feature)
    head.exact_mask_order = exact_order
    cls, masks = head.forward([f.to(DEV) for f in feats], metas)
    torch.cuda.synchronize()
    trace = {}
    head_o.forward(feats, metas, trace=trace)
    e_rel, same = _rel_err(head_o, head, cls, fx["rel"], fx["topk_idx"], trace["query_feat"])
    e_cls = _err(cls["cls"], fx["cls"])
    e_imp = _err(cls["importance"], fx["importance"])
    probe = masks["mask"].flatten()[torch.from_numpy(fx["mask_probe_idx"]).to(DEV)]
    e_mask = _err(probe, fx["mask_probe"])
    print("e2e_full [%s, exact_mask_order=%s] errors: rel %.3e cls %.3e importance %.3e "
          "mask %.3e" % (conv_algo, exact_order, e_rel, e_cls, e_imp, e_mask))
    assert e_rel < 1e-3 and e_cls < 1e-3 and e_imp < 1e-3
    assert e_mask < 1e-3 * max(1.0, float(np.abs(fx["mask_probe"]).max()))
    ok, exact = tie_aware_topk_match(fx["importance"][0], fx["topk_idx"][0],
                                     head._last_plan.topk_idx[0].cpu().numpy(), TIE_TOL_FULL)
    print("%d/100 top-k positions identical" % exact)
    # (the reference's own fp64 evaluation agrees with its fp32 list in no more positions:
    # a few attention-mask bits flip per 800x1333 forward, LABNOTES.md section 3)
    assert ok and exact >= 85
    # size-independent properties at the full size
    first_rel, first_idx = cls["rel"].clone(), head._last_plan.topk_idx.clone()
    again_cls, again_masks = head.forward([f.to(DEV) for f in feats], metas)
    assert torch.equal(again_cls["rel"], first_rel)           # deterministic run to run
    assert torch.equal(head._last_plan.topk_idx, first_idx)
    neg = float((masks["mask"] < 0).float().mean())
    assert abs(neg - float(fx["mask_neg_frac"])) < 1e-3
    res = head.get_bboxes(again_cls, again_masks, metas)
    assert res[0][3].shape == (200, round(H / 2.083), round(W / 2.083))
    assert res[0][7].shape == (100, 57)
    assert float((res[0][7].sum(-1) - 1).abs().max()) < 1e-5


@pytest.mark.parametrize("name,fuse", [("e2e_small_sep", False), ("e2e_full_sep", False),
                                       ("e2e_small_sep", "front"), ("e2e_full_sep", "front")])
def test_e2e_topk_pair_indices_bit_exact_on_separated_fixtures(name, fuse):
    """north_star: "top-k pair indices bit-exact", end to end, at 96x128 (batch 2) and at
    800x1333: strict equality of topk_idx / sub_pos / obj_pos with the reference's, relation
    logits compared directly with the reference's (no re-evaluation), get_bboxes labels
    exact."""
    fx = golden(name)
    _, sd, _ = oracle_head(int(fx["weight_seed"]), overrides_of(fx))
    assert seeded.checksum(sd) == int(fx["weight_crc"])
    H, W, bs, sf = int(fx["height"]), int(fx["width"]), int(fx["batch"]), float(fx["img_scale"])
    feats = seeded.seeded_feats(int(fx["feat_seed"]), bs, H, W)
    assert seeded.checksum(feats) == int(fx["feat_crc"])
    metas = [dict(img_shape=(H, W, 3), scale_factor=[sf] * 4)] * bs
    assert float(fx["min_gap"]) >= 1e-4 and float(fx["min_gap"]) >= 10 * float(fx["fp64_noise"])
    head = _hip_head(sd)
    head.fuse_ppn_front = fuse == "front"   # (k_ppn_front: LDS-staged query tiles, csrc/ppn.hip)
    cls, masks = head.forward([f.to(DEV) for f in feats], metas)
    torch.cuda.synchronize()
    pl = head._last_plan
    imp = cls["importance"].cpu().numpy().reshape(bs, -1)
    ref = fx["importance"].reshape(bs, -1)
    top = np.argsort(-ref, axis=1)[:, :200]
    e_top = float(np.abs(np.take_along_axis(imp - ref, top, 1)).max())
    print("%s: min gap %.3e, GPU score error on the top pairs %.3e (margin %.0f; the "
          "reference's fp32-vs-fp64 error is %.3e)" % (name, float(fx["min_gap"]), e_top,
                                                       float(fx["min_gap"]) / max(e_top, 1e-12),
                                                       float(fx["fp64_noise"])))
    assert np.array_equal(pl.topk_idx.cpu().numpy(), fx["topk_idx"])
    assert np.array_equal(pl.sub_pos.cpu().numpy(), fx["sub_pos"])
    assert np.array_equal(pl.obj_pos.cpu().numpy(), fx["obj_pos"])
    assert 2 * e_top < float(fx["min_gap"])     # order-preserving bound (test_production_gpu._check_head)
    errs = {k: _err(cls[k], fx[k]) for k in ("rel", "cls", "sub", "obj")}
    errs["importance"] = float(np.abs(imp - ref).max()) / max(1.0, float(np.abs(ref).max()))
    probe = masks["mask"].flatten()[torch.from_numpy(fx["mask_probe_idx"]).to(DEV)]
    errs["mask"] = _err(probe, fx["mask_probe"]) / max(1.0, float(np.abs(fx["mask_probe"]).max()))
    print(name, "errors:", errs)
    assert all(e < 1e-3 for e in errs.values()), errs
    res = head.get_bboxes(cls, masks, metas)
    for i, r in enumerate(res):
        assert np.array_equal(r[1].cpu().numpy(), fx["res%d_labels" % i])
        assert _err(r[7], fx["res%d_r_dists" % i]) < 1e-3
        shape = tuple(fx["res%d_masks_shape" % i])
        ref_masks = np.unpackbits(fx["res%d_masks" % i])[:int(np.prod(shape))].reshape(shape)
        got = r[3].cpu().numpy()[fx["res%d_masks_rows" % i]]
        assert got.shape == shape and r[3].shape[0] == 200
        assert (got != ref_masks.astype(bool)).mean() < 1e-3
        assert (r[4].cpu().numpy() != fx["res%d_pan_img" % i]).mean() < 5e-3


@pytest.mark.parametrize("name", ["ppn", "ppn_sep"])
def test_pair_proposal_chain_on_golden_query_features(name):
    """G1 through the HIP path: the reference's last-layer query features -> sub / obj
    MLPs, L2 normalisation, Q x Q cosine matrix, Matrix Learner, top-k (pairnet_head.py:
    322-340) against the reference's recorded importance_raw / importance / indices."""
    fx = golden(name)
    _, sd, _ = oracle_head(int(fx["weight_seed"]), overrides_of(fx))
    head = _hip_head(sd)
    q = torch.from_numpy(fx["query_feat"])                     # (Q, B, 256) seq-first
    B = q.shape[1]
    pl = head._plan(B, [(3, 4), (6, 8), (12, 16)], (24, 32))
    pl.q.copy_(q.transpose(0, 1).reshape(-1, 256).to(DEV))
    head._pair_proposal(pl)
    torch.cuda.synchronize()
    e_raw = _err(pl.imp_raw, fx["importance_raw"])
    scale = max(1.0, float(np.abs(fx["importance"]).max()))
    e_imp = _err(pl.imp, fx["importance"]) / scale
    gap = float(np.min(fx["min_gap"]))
    print("%s: importance_raw err %.2e, importance err %.2e (relative to %.2f), min gap %.2e"
          % (name, e_raw, e_imp, scale, gap))
    assert e_raw < 1e-5 and e_imp < 1e-5
    assert np.array_equal(pl.topk_idx.cpu().numpy(), fx["topk_idx"].reshape(B, -1))
    assert np.array_equal(pl.sub_pos.cpu().numpy(), fx["sub_pos"].reshape(B, -1))
    assert np.array_equal(pl.obj_pos.cpu().numpy(), fx["obj_pos"].reshape(B, -1))
    # the gathered pair features are exactly the selected query rows
    pair = pl.pair.view(B, 200, 256).cpu()
    want = torch.stack([q[:, b][torch.from_numpy(np.concatenate(
        [fx["sub_pos"].reshape(B, -1)[b], fx["obj_pos"].reshape(B, -1)[b]]))] for b in range(B)])
    assert torch.equal(pair, want)


def test_relation_decoder_on_golden_pair_features():
    """G3 through the HIP path: pair features -> six Relation Fusion layers -> relation
    logits (pairnet_head.py:353-378) against the reference's recorded rel_preds."""
    fx = golden("reldec")
    _, sd, _ = oracle_head(int(fx["weight_seed"]))
    head = _hip_head(sd)
    pair = torch.from_numpy(fx["pair_feat"])                   # (2R, B, 256) seq-first
    B = pair.shape[1]
    pl = head._plan(B, [(3, 4), (6, 8), (12, 16)], (24, 32))
    pl.pair.copy_(pair.transpose(0, 1).reshape(-1, 256).to(DEV))
    for t in (pl.cls, pl.MP):
        t.zero_()
    pl.sub_pos.zero_()
    pl.obj_pos.zero_()
    head._relation_decoder(pl)
    torch.cuda.synchronize()
    e = _err(pl.rel, fx["rel_preds"])
    print("reldec: rel_preds err %.2e" % e)
    assert e < 1e-4


@pytest.mark.parametrize("exact_mask_order", [False, True, "full"])
def test_against_oracle_other_seed_and_batch_consistency(exact_mask_order):
    head_o, sd, _ = oracle_head(1234)
    H, W = 64, 96
    feats = seeded.seeded_feats(99, 2, H, W)
    feats = [torch.cat([f[:1], f[:1], f[1:]], 0) for f in feats]      # images 0 and 1 identical
    metas = [dict(img_shape=(H, W, 3), scale_factor=[1.5] * 4)] * 3
    trace = {}
    ref_cls, ref_masks = head_o.forward(feats, metas, trace=trace)
    head = _hip_head(sd)
    head.exact_mask_order = exact_mask_order
    cls, masks = head.forward([f.to(DEV) for f in feats], metas)
    torch.cuda.synchronize()
    for k in ("cls", "importance"):
        e = _err(cls[k], ref_cls[k])
        print(k, e)
        assert e < 1e-3, k
    e, same = _rel_err(head_o, head, cls, ref_cls["rel"], trace["topk_idx"], trace["query_feat"])
    print("rel", e, "top-k identical:", same)
    assert e < 1e-3
    assert _err(masks["mask"], ref_masks["mask"]) < 1e-3 * max(1.0, float(ref_masks["mask"].abs().max()))
    assert _err(head._last_plan.q.view(3, 100, 256), trace["query_feat"].transpose(0, 1)) < 1e-3
    # no cross-image op on the path: identical images give identical rows
    for d in (cls, masks):
        for k, v in d.items():
            assert torch.equal(v[0], v[1]), k


def test_forward_head_signature_and_values():
    head_o, sd, _ = oracle_head(4321)
    head = _hip_head(sd)
    g = torch.Generator().manual_seed(5)
    dec = torch.randn(100, 2, 256, generator=g)
    mf = torch.randn(2, 256, 24, 32, generator=g)
    ref = head_o.forward_head(dec, mf, (6, 8))
    out = head.forward_head(dec.to(DEV), mf.to(DEV), (6, 8))
    assert _err(out[0], ref[0]) < 1e-4 and _err(out[1], ref[1]) < 1e-3
    assert out[2].shape == ref[2].shape == (16, 100, 48) and out[2].dtype == torch.bool
    assert float((out[2].cpu() != ref[2]).float().mean()) < 1e-3


def test_graph_replay_and_two_stream_pipeline_are_bitwise_the_eager_path():
    """hipGraph replay per stage and the 2-stream pipeline reorder launches, never
    arithmetic: every output must equal the eager single-stream call bit for bit."""
    from pairnet_amd import PipelinedHead
    head_o, sd, _ = oracle_head(77)
    head = _hip_head(sd)
    H, W = 64, 96
    batches = [[f.to(DEV) for f in seeded.seeded_feats(200 + i, 2, H, W)] for i in range(5)]
    metas = [dict(img_shape=(H, W, 3), scale_factor=[1.5] * 4)] * 2
    names = ("bboxes", "labels", "rel_pairs", "masks", "pan", "r_scores", "r_labels", "r_dists")

    def snap(res):
        return [[t.cpu().clone() if isinstance(t, torch.Tensor) else t for t in r] for r in res]
    eager = [snap(head.simple_test_bboxes(b, metas)) for b in batches]
    head.use_graphs = True
    for rep in range(3):                                   # eager -> capture -> replay
        got = [snap(head.simple_test_bboxes(b, metas)) for b in batches]
        for e, g in zip(eager, got):
            for re_, rg in zip(e, g):
                for n, x, y in zip(names, re_, rg):
                    assert torch.equal(x, y), (rep, n)
    for depth, a_streams in ((2, 1), (3, 1), (4, 2), (5, 2)):
        eng = PipelinedHead(head, depth=depth, a_streams=a_streams)
        for rep in range(3):
            outs = []
            for b in batches:
                r = eng.submit(b, metas)
                if r is not None:
                    outs.append(snap(r))
            outs.extend(snap(r) for r in eng.flush())
            assert len(outs) == len(batches)
            for e, g in zip(eager, outs):
                for re_, rg in zip(e, g):
                    for n, x, y in zip(names, re_, rg):
                        assert torch.equal(x, y), (depth, a_streams, rep, n)


def test_graph_reads_reused_feature_buffers_in_place():
    """A caller that hands over the SAME buffers every step (new contents written in place)
    gets stage A's graph captured on those buffers -- no staging copy -- and still sees
    every step's own features; a caller that switches buffers is served eagerly (no private
    copies per shape), and the graph comes back once its buffers repeat."""
    _, sd, _ = oracle_head(77)
    head = _hip_head(sd)
    H, W = 64, 96
    metas = [dict(img_shape=(H, W, 3), scale_factor=[1.0] * 4)] * 2
    batches = [[f.to(DEV) for f in seeded.seeded_feats(300 + i, 2, H, W)] for i in range(4)]
    eager = []
    for b in batches:
        cls, _ = head.forward(b, metas)
        eager.append(cls["rel"].clone())
    head.use_graphs = True
    bufs = [torch.empty_like(f) for f in batches[0]]
    for rep in range(2):
        for b, want in zip(batches, eager):
            for dst, src in zip(bufs, b):
                dst.copy_(src)
            cls, _ = head.forward(bufs, metas)
            assert torch.equal(cls["rel"], want)
    pl = head._last_plan
    assert pl.graph_a is not None and pl.static_ptrs == tuple(f.data_ptr() for f in bufs)
    mem = torch.cuda.memory_allocated()
    for b, want in zip(batches, eager):                    # other buffers every call: eager
        cls, _ = head.forward(b, metas)
        assert torch.equal(cls["rel"], want)
    assert head._last_plan is pl and pl.graph_a is None
    assert torch.cuda.memory_allocated() == mem            # ... and nothing was staged
    for rep in range(2):                                   # the same buffers again: re-captured
        for dst, src in zip(bufs, batches[2]):
            dst.copy_(src)
        cls, _ = head.forward(bufs, metas)
        assert torch.equal(cls["rel"], eager[2])
    assert pl.graph_a is not None


def test_in_place_weight_updates_invalidate_packed_weights_constants_and_graphs():
    """plans cache weight-derived state (packed / folded weights, the mask
    embedding of the initial queries, position tables with the level embeddings folded in,
    captured graphs).  A parameter written IN PLACE -- what an optimizer step does -- must be
    seen by the next forward: the result equals a fresh head loaded with the updated state."""
    _, sd, _ = oracle_head(77)
    head = _hip_head(sd)
    head.use_graphs = True
    H, W = 64, 96
    metas = [dict(img_shape=(H, W, 3), scale_factor=[1.0] * 4)]
    feats = [f.to(DEV) for f in seeded.seeded_feats(301, 1, H, W)]
    for _ in range(3):                                      # eager, capture, replay
        cls, _ = head.forward(feats, metas)
    before = cls["rel"].clone()
    assert head._last_plan.graph_b is not None
    params = dict(zip(head.state_dict().keys(), head.parameters()))
    with torch.no_grad():
        params["level_embed.weight"].mul_(1.5)              # folded into the position tables
        params["query_feat.weight"].add_(0.25)              # the constants q0 / me0
        params["rel_cls_embed.bias"].add_(1.0)              # a packed weight
    for _ in range(3):
        cls, _ = head.forward(feats, metas)
    after = cls["rel"].clone()
    assert not torch.equal(before, after)
    fresh = _hip_head(head.state_dict())
    want, _ = fresh.forward(feats, metas)
    assert torch.equal(after, want["rel"])


@pytest.mark.parametrize("B,H,W", [(2, 64, 80)])
def test_swin_l_200_query_configuration(B, H, W):
    """BASELINE.json configs[3]: Swin-L channel widths, 200 object queries (200x200
    importance matrix, top-k over 40 000), batch 2 at a small size against the CPU oracle on
    UNSEPARATED seeded weights (hence the tie-aware comparison).  The production size is
    covered strictly since round 4: tests/test_production_gpu.py, fixture
    `e2e_image_swinl_full` (true Swin-L -> reference CrossHead2 at 800 x 1333, index-exact)."""
    from collections import OrderedDict
    from oracle.head import OracleCrossHead2
    from pairnet_amd import CrossHead2, pairnet_head_cfg
    cfg = pairnet_head_cfg(in_channels=(192, 384, 768, 1536), num_obj_query=200)
    cfg.pop("type")
    head_o = OracleCrossHead2(**cfg).eval()
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in head_o.state_dict().items())
    sd = seeded.seeded_state_dict(shapes, 4242)
    head_o.load_state_dict(sd)
    head = CrossHead2(**cfg)
    head.load_state_dict(sd)
    head.to(DEV)
    full = H >= 800
    feats = seeded.seeded_feats(17, B, H, W, channels=(192, 384, 768, 1536))
    sf = 2.083 if full else 1.0
    metas = [dict(img_shape=(H, W, 3), scale_factor=[sf] * 4)] * B
    trace = {}
    with torch.no_grad():
        ref_cls, ref_masks = head_o.forward(feats, metas, trace=trace)
    cls, masks = head.forward([f.to(DEV) for f in feats], metas)
    assert cls["importance"].shape == (B, 200, 200) and cls["rel"].shape == (B, 100, 56)
    e_cls, e_imp = _err(cls["cls"], ref_cls["cls"]), _err(cls["importance"], ref_cls["importance"])
    e, same = _rel_err(head_o, head, cls, ref_cls["rel"], trace["topk_idx"], trace["query_feat"])
    print("Swin-L widths, Q = 200, %dx%d: cls %.2e importance %.2e rel %.2e" % (H, W, e_cls, e_imp, e))
    assert e_cls < 1e-3 and e_imp < 1e-3 and e < 1e-3
    for b in range(B):
        ok, exact = tie_aware_topk_match(ref_cls["importance"][b].numpy(),
                                         trace["topk_idx"][b].numpy(),
                                         head._last_plan.topk_idx[b].cpu().numpy(),
                                         TIE_TOL_FULL if full else TIE_TOL_SMALL)
        print("image %d: %d/100 top-k positions identical" % (b, exact))
        assert ok
    res = head.get_bboxes(cls, masks, metas)
    h0, w0 = round(H / sf), round(W / sf)
    assert res[0][3].shape == (200, h0, w0) and res[0][1].shape == (200,)


def _crafted_postproc_inputs(seed, Q=100, h=24, w=32, empty=False):
    """Confident class logits for a handful of queries (stuff duplicates, the excluded
    class 132, one query whose mask wins only a few pixels) so that every branch of
    _get_bboxes_single runs: keep filter, stuff merging, the area <= 4 drop-and-redo."""
    g = torch.Generator().manual_seed(seed)
    all_cls = torch.randn(Q, 134, generator=g) * 0.3
    if not empty:
        conf = {3: 17, 8: 100, 15: 100, 21: 132, 30: 5, 41: 90, 55: 90, 56: 90, 70: 2, 88: 119}
        for q, lab in conf.items():
            all_cls[q, lab] += 12.0
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    all_masks = torch.randn(Q, h, w, generator=g) * 0.5
    for q in range(Q):   # smooth blobs, different centre per query
        cy, cx = (q * 7) % h, (q * 13) % w
        all_masks[q] += 6.0 * torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / (2 * (2.0 + q % 5) ** 2))
    all_masks[30] = -5.0
    all_masks[30, 5, 5] = 30.0          # wins ~1 low-res pixel -> area <= 4 after upsampling? no:
    all_masks[70] = -20.0               # never wins: area 0 -> dropped, forces a second pass
    s_cls, o_cls = torch.randn(100, 134, generator=g), torch.randn(100, 134, generator=g)
    r_cls = torch.randn(100, 56, generator=g)
    s_seg, o_seg = torch.randn(100, h, w, generator=g), torch.randn(100, h, w, generator=g)
    return all_masks, all_cls, s_cls, o_cls, r_cls, s_seg, o_seg


@pytest.mark.parametrize("case", ["rich", "empty"])
def test_device_postprocessing_equals_reference_loop(case):
    """get_bboxes on crafted inputs vs the oracle's restatement of the reference's host
    loop (pairnet_head.py:788-924): labels, r_dists, masks, pan_img, including stuff
    merging and the small-area re-run.  No host sync inside the device version."""
    head_o, sd, _ = oracle_head(5)
    head = _hip_head(sd)
    args = _crafted_postproc_inputs(11, empty=(case == "empty"))
    img_shape, sf = (48, 64, 3), [1.0, 1.0, 1.0, 1.0]
    ref = head_o._get_bboxes_single(*args, img_shape, sf)
    got = head._get_bboxes_single(*[a.to(DEV) for a in args], img_shape, sf)
    st = head.panoptic_status()[0]
    print(st)
    assert torch.equal(got[1].cpu(), ref[1])                       # labels
    assert _err(got[7], ref[7]) < 1e-6                              # r_dists
    assert float((got[3].cpu() != ref[3]).float().mean()) < 1e-3    # masks (|logit|~0 pixels)
    assert torch.equal(got[4].cpu(), ref[4])                        # pan_img, exact
    if case == "rich":
        assert st["nkeep"] == 9 and st["rounds"] >= 1
        assert len(torch.unique(ref[4])) >= 4
    else:
        assert st["nkeep"] == 0 and int(got[4].min()) == 1 and int(got[4].max()) == 1


def test_panoptic_loop_converges_like_the_reference_or_raises(monkeypatch):
    """pairnet_head.py:893-905 loops until no segment of area <= 4 is left.  (a) A case
    that needs the maximum the reference can need -- the merged-stuff round, a round in
    which a duplicate that only existed inside the merge falls out, and the final round
    (dropping segments only ever grows the others, so nothing can fall out later) -- with
    only ONE round enqueued up front: `panoptic_status` continues the loop to convergence
    and the map equals the reference's.  (b) Every segment filtered: the reference fails
    (:882), and so does `panoptic_status`."""
    from pairnet_amd import hip
    head_o, sd, _ = oracle_head(5)
    head = _hip_head(sd)
    args = list(_crafted_postproc_inputs(11))
    all_masks, all_cls = args[0], args[1]
    # queries 41 / 55 / 56 share stuff class 90 and are merged into 41 in round 0 (55 / 56
    # then have area 0 and are dropped).  41 itself only owns three pixels: it survives round
    # 0 on the merged area, loses it in round 1 (the duplicates' pixels go to whoever is next
    # best there) and is dropped then -- the longest chain the reference's loop can take
    all_masks[41] = -30.0
    all_masks[41, 10, 10:13] = 40.0
    img_shape, sf = (24, 32, 3), [1.0, 1.0, 1.0, 1.0]            # no upsampling
    ref = head_o._get_bboxes_single(*args, img_shape, sf)
    monkeypatch.setattr(hip, "PAN_ROUNDS", 1)
    got = head._get_bboxes_single(*[a.to(DEV) for a in args], img_shape, sf)
    st = head.panoptic_status()[0]
    print(st)
    assert st["rounds"] == 2                                        # two rounds dropped something
    assert torch.equal(got[4].cpu(), ref[4])
    monkeypatch.setattr(hip, "PAN_ROUNDS", 4)
    got = head._get_bboxes_single(*[a.to(DEV) for a in args], img_shape, sf)
    assert head.panoptic_status()[0]["rounds"] == 2 and torch.equal(got[4].cpu(), ref[4])
    # (b) a 2 x 2 output with three confident queries: every area is <= 4
    tiny = (2, 2, 3)
    with pytest.raises((IndexError, RuntimeError)):
        head_o._get_bboxes_single(*args, tiny, sf)
    head._get_bboxes_single(*[a.to(DEV) for a in args], tiny, sf)
    with pytest.raises(IndexError):
        head.panoptic_status()


def test_evaluator_feed_mask_iou_counts_are_exact():
    from pairnet_amd import hip
    g = torch.Generator().manual_seed(1)
    P, G, H, W = 37, 11, 61, 83                  # HW not a multiple of 64
    pred = torch.rand(P, H, W, generator=g) > 0.6
    gt = torch.rand(G, H, W, generator=g) > 0.7
    gt[3] = False
    pred[5] = gt[2]
    nw = (H * W + 63) // 64
    pw = torch.empty(P, nw, device=DEV, dtype=torch.int64)
    gw = torch.empty(G, nw, device=DEV, dtype=torch.int64)
    hip.pack_mask_bits(pred.to(DEV).view(torch.uint8), pw, P, H * W)
    hip.pack_mask_bits(gt.to(DEV).view(torch.uint8), gw, G, H * W)
    bits = np.unpackbits(pw.cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :H * W]
    assert np.array_equal(bits.astype(bool), pred.flatten(1).numpy())
    inter = torch.empty(P, G, device=DEV, dtype=torch.int32)
    ap = torch.empty(P, device=DEV, dtype=torch.int32)
    ag = torch.empty(G, device=DEV, dtype=torch.int32)
    hip.mask_iou_counts(pw, P, gw, G, nw, inter, ap, ag)
    pf, gf = pred.flatten(1).long(), gt.flatten(1).long()
    assert torch.equal(inter.cpu().long(), pf @ gf.t())
    assert torch.equal(ap.cpu().long(), pf.sum(1)) and torch.equal(ag.cpu().long(), gf.sum(1))
    # mask_iou (sgg_metrics.py:1374-1380) from the counts, in float64 like numpy
    i, a, b = inter.cpu().double(), ap.cpu().double()[:, None], ag.cpu().double()[None]
    iou = i / (a + b - i)
    assert float(iou[5, 2]) == 1.0


# ---- G4 / G5 fixtures: forward_head and one masked decoder layer --------
def test_forward_head_against_reference_golden():
    fx = golden("fwdhead")
    _, sd, crc = oracle_head(int(fx["weight_seed"]))
    assert crc == int(fx["weight_crc"])
    head = _hip_head(sd)
    cls, mask, attn = head.forward_head(torch.from_numpy(fx["decoder_out"]).to(DEV),
                                        torch.from_numpy(fx["mask_feature"]).to(DEV), (8, 12))
    torch.cuda.synchronize()
    assert _err(cls, fx["cls_pred"]) < 1e-4
    assert _err(mask, fx["mask_pred"]) < 1e-3 * max(1.0, float(np.abs(fx["mask_pred"]).max()))
    shape = tuple(fx["attn_shape"])
    ref_bits = np.unpackbits(fx["attn_mask"])[:int(np.prod(shape))].reshape(shape).astype(bool)
    got = attn.cpu().numpy()
    assert got.shape == shape and got.dtype == np.bool_
    # the threshold is a discrete decision: positions whose reference logit is
    # within 1e-4 of it are exempt, every other bit must agree
    sure = np.abs(fx["resized_logits"]).reshape(2, 1, 100, 96) > 1e-4
    sure = np.broadcast_to(sure, (2, 8, 100, 96)).reshape(shape)
    assert np.array_equal(got[sure], ref_bits[sure])
    assert sure.mean() > 0.999


def test_masked_decoder_layer_against_reference_golden():
    """One cross-attn -> self-attn -> FFN layer (transformer_decoder.layers[0]) on the
    reference's own inputs, through the kernels `_layer` strings together."""
    from pairnet_amd import hip
    fx = golden("declayer")
    _, sd, crc = oracle_head(int(fx["weight_seed"]))
    assert crc == int(fx["weight_crc"])
    head = _hip_head(sd)
    head._pack()
    w = head.w
    Q, B, K = 100, 2, 96
    E = lambda *s: torch.empty(*s, device=DEV)
    bf = lambda a: torch.from_numpy(a).permute(1, 0, 2).contiguous().to(DEV)   # (L,B,C)->(B,L,C)
    x = bf(fx["query"]).view(B * Q, 256).clone()
    xpos = torch.from_numpy(fx["query_pos"][:, 0]).contiguous().to(DEV)
    mem, kpos = bf(fx["memory"]), torch.from_numpy(fx["key_pos"][:, 0]).contiguous().to(DEV)
    a0 = "transformer_decoder.layers.0.attentions.0.attn."
    Kp, Vp = E(B, K, 256), E(B, K, 256)
    hip.linear(mem.view(-1, 256), w[a0 + "in_proj_weight"][256:512], w[a0 + "in_proj_bias"][256:512],
               Kp.view(-1, 256), aadd=kpos)
    hip.linear(mem.view(-1, 256), w[a0 + "in_proj_weight"][512:], w[a0 + "in_proj_bias"][512:],
               Vp.view(-1, 256))
    shape = tuple(fx["mask_shape"])
    mask = np.unpackbits(fx["mask"])[:int(np.prod(shape))].reshape(shape).astype(bool)
    logits = torch.from_numpy(np.where(mask, -1.0, 1.0).astype(np.float32)).to(DEV).view(B * Q, K)
    bits = torch.empty(B * Q * ((K + 31) // 32), device=DEV, dtype=torch.int32)
    rowall = torch.empty(B * Q, device=DEV, dtype=torch.int32)
    hip.mask_pack(logits, bits, rowall, B * Q, K)
    scr = E(max(hip.attn_scratch_floats(B, Q, K), hip.attn_scratch_floats(B, Q, Q)))
    hbuf = E(hip.ffn_scratch_floats(B * Q, head.dec_ffn))
    x1, x2, y, Qp, att = (E(B * Q, 256) for _ in range(5))
    VQK = E(B * Q, 768)
    head._layer("transformer_decoder.layers.0.", x, xpos, x1, x2, y, Qp, VQK, att, hbuf, Kp, 256,
                Vp, 256, K, B, Q, bits, rowall, scr, head.dec_ffn)
    torch.cuda.synchronize()
    got = x.view(B, Q, 256).permute(1, 0, 2)
    assert _err(got, fx["out"]) < 1e-4


def test_get_bboxes_with_different_image_shapes_in_one_batch():
    """Two images of one batch with their own `img_shape` / `scale_factor` (a padded batch):
    the reference resizes every image's masks to round(img_shape / scale_factor) of ITS meta
    (pairnet_head.py:803-806, no crop -- kept); the device post-processing against the oracle's
    restatement of the host loop on the same head outputs, with graph replay on and off."""
    head_o, sd, _ = oracle_head(7)
    H, W = 96, 128
    feats = seeded.seeded_feats(21, 2, H, W)
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0] * 4),
             dict(img_shape=(80, 100, 3), scale_factor=[1.5, 1.25, 1.5, 1.25])]
    for graphs in (False, True):
        head = _hip_head(sd)
        head.use_graphs = graphs
        for rep in range(3 if graphs else 1):
            cls, masks = head.forward([f.to(DEV) for f in feats], metas)
            res = head.get_bboxes(cls, masks, metas)
        torch.cuda.synchronize()
        ref = head_o.get_bboxes({k: v.cpu() for k, v in cls.items()},
                                {k: v.cpu() for k, v in masks.items()}, metas)
        assert res[0][3].shape == (200, 48, 64) and res[1][3].shape == (200, 64, 67)
        for r, o in zip(res, ref):
            assert torch.equal(r[1].cpu(), o[1])                                  # labels
            assert _err(r[7], o[7]) < 1e-6                                        # r_dists
            assert r[3].shape == o[3].shape and r[4].shape == o[4].shape
            assert float((r[3].cpu() != o[3]).float().mean()) < 1e-3             # masks
            assert float((r[4].cpu() != o[4]).float().mean()) < 5e-3             # pan_img
            assert np.array_equal(r[2].numpy() if not r[2].is_cuda else r[2].cpu().numpy(),
                                  np.asarray(o[2]))


def test_stage_a_graph_table_evicts_the_least_recently_used_entry():
    """a plan keeps at most STAGE_A_GRAPHS stage-A graphs, one per set of caller feature
    buffers.  When all of them own a graph and the caller arrives with yet another buffer set (a
    backbone whose arena grew), the least recently used entry goes -- with it the reference that
    kept the old buffers alive -- instead of the newcomer being served eagerly for ever; results
    stay the eager ones throughout."""
    from pairnet_amd import CrossHead2
    head = CrossHead2(**head_cfg())
    head.init_weights(seed=2)
    head.to(DEV)
    H, W = 64, 96
    g = torch.Generator().manual_seed(8)
    feats = [torch.randn(1, c, H // s, W // s, generator=g).to(DEV)
             for c, s in zip((256, 512, 1024, 2048), (4, 8, 16, 32))]
    metas = [dict(img_shape=(H, W, 3), scale_factor=[1.0] * 4)]
    head.use_graphs = False
    ref = head.forward(feats, metas)[0]["rel"].clone()
    head.use_graphs = True
    sets = [[f.clone() for f in feats] for _ in range(head.STAGE_A_GRAPHS + 2)]
    key = lambda s: tuple(f.data_ptr() for f in s)
    for s in sets[:head.STAGE_A_GRAPHS] * 2 + [sets[-2]] * 2 + [sets[0]] * 2 + [sets[-1]] * 2:
        out = head.forward(s, metas)[0]["rel"]
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
    pl = head._last_plan
    assert len(pl.graphs_a) == head.STAGE_A_GRAPHS
    assert all(e["graph"] is not None for e in pl.graphs_a.values())
    # the two newcomers and the re-admitted first set are in, the least recently used ones are out
    assert key(sets[-1]) in pl.graphs_a and key(sets[-2]) in pl.graphs_a and key(sets[0]) in pl.graphs_a
    assert key(sets[1]) not in pl.graphs_a and key(sets[2]) not in pl.graphs_a
