"""GPU parity of the sibling head CrossHeadBBox (pairnet_bbox_head.py) and its ChannelMapper neck:
the glue kernels of csrc/detr.hip against torch restatements of the reference's lines, the neck
and the head against the oracle (oracle/deformable_detr.py, oracle/bbox_head.py -- pinned to the
reference class and to transformers' Deformable DETR in tests/test_oracle.py) and against the
golden fixtures recorded from the reference class (oracle/make_golden.py::gen_bbox)."""
import math
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import golden, overrides_of
from oracle import seeded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip(built_lib):
    import importlib
    return importlib.import_module("pairnet_amd.hip")


def _err(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(np.asarray(b)).double()).abs().max())


def _cfgs():
    import pairnet_amd as P
    cfg = {k: v for k, v in P.bbox_head_cfg().items() if k != "type"}
    ncfg = {k: v for k, v in P.channel_mapper_cfg().items() if k != "type"}
    return P, cfg, ncfg


def _oracles(weight_seed, neck_seed, ops=None):
    from oracle.bbox_head import OracleCrossHeadBBox
    from oracle.deformable_detr import ChannelMapper
    _, cfg, ncfg = _cfgs()
    head, neck = OracleCrossHeadBBox(**cfg).eval(), ChannelMapper(**ncfg).eval()
    sd = seeded.seeded_state_dict(
        OrderedDict((k, tuple(v.shape)) for k, v in head.state_dict().items()), weight_seed)
    nsd = seeded.seeded_state_dict(
        OrderedDict((k, tuple(v.shape)) for k, v in neck.state_dict().items()), neck_seed)
    crc = (seeded.checksum(sd), seeded.checksum(nsd))
    seeded.apply_ops(sd, ops or {})
    head.load_state_dict(sd, strict=True)
    neck.load_state_dict(nsd, strict=True)
    return head, neck, sd, nsd, crc


def _hip_models(sd, nsd):
    P, cfg, ncfg = _cfgs()
    neck = P.ChannelMapper(**ncfg).to(DEV)
    neck.load_state_dict(nsd)
    head = P.CrossHeadBBox(**cfg).to(DEV)
    head.load_state_dict(sd)
    return head, neck


def _feats(seeds, H, W, smooth=0):
    per = [(seeded.smooth_feats(int(s), 1, H, W, smooth) if smooth
            else seeded.seeded_feats(int(s), 1, H, W))[1:] for s in seeds]
    return [torch.cat([p[l] for p in per], 0) for l in range(3)]


# ---------------------------------------------------------------- glue kernels
def test_box_pos_embed_and_refine(hip):
    g = torch.Generator().manual_seed(3)
    u = torch.randn(700, 4, generator=g) * 3
    u[5] = float("inf")                      # an invalid proposal's logits
    ref, emb = torch.empty(700, 4, device=DEV), torch.empty(700, 512, device=DEV)
    hip.box_pos_embed(u.to(DEV), ref, emb, 700)
    from oracle.deformable_detr import DeformableDetrTransformer as T, inverse_sigmoid
    assert _err(ref, u.sigmoid()) < 1e-6
    assert _err(emb, T.get_proposal_pos_embed(u[None])[0]) < 2e-5
    delta = torch.randn(700, 4, generator=g)
    r = torch.rand(700, 4, generator=g)
    r[0], r[1] = 0.0, 1.0                    # the eps clamps of inverse_sigmoid
    out = torch.empty(700, 4, device=DEV)
    hip.box_refine(delta.to(DEV), r.to(DEV), out, 700)
    assert _err(out, (delta + inverse_sigmoid(r)).sigmoid()) < 1e-6


def test_box_sampling_plus_msda_loc_is_mmcv_attention_on_reference_boxes(hip):
    """pn_box_sampling_f32 + pn_msda_loc_f32 against the oracle's MultiScaleDeformableAttention
    core with 4-d reference points."""
    from oracle import layers as L
    g = torch.Generator().manual_seed(4)
    B, Nq, shapes = 2, 37, [(12, 16), (6, 8), (3, 4), (2, 2)]
    N = sum(h * w for h, w in shapes)
    value = torch.randn(B, N, 256, generator=g)
    offaw = torch.cat([torch.randn(B * Nq, 256, generator=g) * 3, torch.randn(B * Nq, 128, generator=g)], 1)
    ref = torch.rand(B * Nq, 4, generator=g) * torch.tensor([1, 1, 0.5, 0.5])
    off = offaw[:, :256].view(B, Nq, 8, 4, 4, 2)
    aw = offaw[:, 256:].view(B, Nq, 8, 16).softmax(-1).view(B, Nq, 8, 4, 4)
    r = ref.view(B, Nq, 1, 4)[:, :, None].expand(B, Nq, 1, 4, 4)[:, :, :, :, None]   # (B,Nq,1,L,1,4)
    loc = r[..., :2] + off / 4 * r[..., 2:] * 0.5
    want = L.msda_core(value.view(B, N, 8, 32), shapes, loc, aw)
    d = lambda t: t.contiguous().to(DEV)
    loc_d, aw_d = torch.empty(B * Nq, 8, 4, 4, 2, device=DEV), torch.empty(B * Nq, 8, 4, 4, device=DEV)
    hip.box_sampling(d(offaw), 384, d(ref), loc_d, aw_d, B * Nq, 4)
    assert _err(loc_d.view(B, Nq, 8, 4, 4, 2), loc) < 1e-6 and _err(aw_d.view(B, Nq, 8, 4, 4), aw) < 1e-6
    out = torch.empty(B * Nq, 256, device=DEV)
    starts = [0, 192, 240, 252]
    hip.msda_loc(d(value), 256, torch.tensor(shapes, device=DEV), torch.tensor(starts, device=DEV),
                 loc_d, aw_d, out, B, N, Nq, 4)
    assert _err(out.view(B, Nq, 256), want) < 2e-5


def test_query_score_topk_strided_zero_rows_sigmoid(hip):
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(2, 300, 150, generator=g) * 2
    score = torch.empty(2, 300, device=DEV)
    hip.query_score(logits.to(DEV), score, 2, 300, 150)
    want = torch.softmax(logits, dim=1).max(-1).values          # pairnet_bbox_head.py:252-254
    assert _err(score, want) < 1e-7
    # proposal selection: the 300 best of column 0 of [B, n, 150]
    n = 22223
    enc = torch.randn(2, n, 150, generator=g)
    idx, q, r = (torch.empty(2, 300, device=DEV, dtype=torch.int64) for _ in range(3))
    hip.topk_strided(enc.to(DEV), 150, n * 150, idx, q, r, 2, n, 1, 300)
    assert torch.equal(idx.cpu(), torch.topk(enc[..., 0], 300, dim=1)[1])
    x = torch.randn(2, 50, 256, generator=g)
    valid = (torch.rand(50, generator=g) > 0.3).to(torch.uint8)
    out = torch.empty(2, 50, 256, device=DEV)
    hip.zero_rows(x.to(DEV), valid.to(DEV), out, 2, 50, 256)
    assert torch.equal(out.cpu(), x * valid.view(1, 50, 1))
    y = torch.tensor([0.3, -4.0, float("inf"), 20.0, -float("inf")])
    o = torch.empty(5, device=DEV)
    hip.sigmoid(y.to(DEV), o)
    assert _err(o, y.sigmoid()) < 1e-7


def test_sine_encoding_with_offset(hip):
    from oracle import layers as L
    pe = L.SinePositionalEncoding(128, normalize=True, offset=-0.5)
    want = pe(torch.zeros(1, 13, 21, dtype=torch.bool))[0].permute(1, 2, 0).reshape(-1, 256)
    out = torch.empty(13 * 21, 256, device=DEV)
    add = torch.randn(256)
    hip.sine_pe(out, add.to(DEV), 13, 21, offset=-0.5)
    assert _err(out, want + add) < 2e-5


def test_proposals_match_the_trunk_restatement():
    """CrossHeadBBox.proposals (host, per shape) == gen_encoder_output_proposals on an unpadded
    batch, bit for bit."""
    import pairnet_amd as P
    from oracle.deformable_detr import DeformableDetrTransformer
    shapes = [(100, 167), (50, 84), (25, 42), (13, 21)]
    prop, valid = P.CrossHeadBBox.proposals(shapes)
    n = prop.shape[0]
    t = DeformableDetrTransformer.__new__(DeformableDetrTransformer)
    torch.nn.Module.__init__(t)
    t.enc_output, t.enc_output_norm = torch.nn.Identity(), torch.nn.Identity()
    mem = torch.ones(1, n, 4)
    om, want = t.gen_encoder_output_proposals(mem, torch.zeros(1, n, dtype=torch.bool), shapes)
    assert torch.equal(prop, want[0]) and torch.equal(valid, om[0, :, 0] == 1)


# ---------------------------------------------------------------- neck
@pytest.mark.parametrize("fmt", ["nchw", "channels_last"])
def test_channel_mapper_matches_oracle(built_lib, fmt):
    _, oneck, _, nsd, _ = _oracles(11, 12)
    P, _, ncfg = _cfgs()
    neck = P.ChannelMapper(**ncfg).to(DEV)
    neck.load_state_dict(nsd)
    feats = _feats([7, 8], 160, 192)
    ins = [f.to(DEV) if fmt == "nchw" else f.to(DEV).contiguous(memory_format=torch.channels_last)
           for f in feats]
    outs = neck(ins)
    with torch.no_grad():
        want = oneck(feats)
    assert len(outs) == 4 and [tuple(o.shape) for o in outs] == [tuple(w.shape) for w in want]
    assert max(_err(o, w) for o, w in zip(outs, want)) < 2e-5


# ---------------------------------------------------------------- head
def _run_fixture(name):
    fx = golden(name)
    ops = overrides_of(fx)
    ohead, oneck, sd, nsd, crc = _oracles(int(fx["weight_seed"]), int(fx["neck_seed"]), ops)
    assert crc == (int(fx["weight_crc"]), int(fx["neck_crc"]))
    H, W, bs = int(fx["height"]), int(fx["width"]), int(fx["batch"])
    feats = _feats(fx["feat_seeds"], H, W, int(fx["feat_smooth"]) if "feat_smooth" in fx.files else 0)
    assert seeded.checksum(feats) == int(fx["feat_crc"])
    sf = [float(v) for v in fx["img_scale"]]
    metas = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=sf)] * bs
    head, neck = _hip_models(sd, nsd)
    cls, box = head(neck([f.to(DEV) for f in feats]), metas)
    torch.cuda.synchronize()
    return fx, head, cls, box, metas


@pytest.mark.parametrize("name", ["bbox_small", "bbox_full"])
def test_bbox_head_against_reference_golden(built_lib, name):
    """The three index selections of the path (300 proposals as a set -- their order only
    permutes the decoder queries --, the 100 kept queries in rank order, the 100 pairs in rank
    order) are EXACTLY the reference's; logits / boxes within 1e-3; get_bboxes labels exact."""
    fx, head, cls, box, metas = _run_fixture(name)
    pl = head._last_plan
    bs = int(fx["batch"])
    for k in ("prop", "keep", "pair"):
        assert float(fx[k + "_gap"]) >= 5 * float(fx[k + "_noise"])
    got_prop = pl.top_idx.cpu().numpy()
    assert np.array_equal(np.sort(got_prop, -1), np.sort(fx["proposals"], -1))
    # a decoder query is identified by its proposal's token index (the order inside the
    # proposal set only permutes the queries): compare rankings token by token
    SN = pl.SN
    by_token = np.full((bs, SN), np.nan, np.float32)
    np.put_along_axis(by_token, got_prop, pl.qscore.cpu().numpy(), 1)
    e_keep = float(np.abs(np.take_along_axis(by_token, fx["proposals"], 1) - fx["query_score"]).max())
    imp = cls["importance"].cpu().numpy().reshape(bs, -1)
    ref = fx["cls_importance"].reshape(bs, -1)
    top = np.argsort(-ref, axis=1)[:, :200]
    e_pair = float(np.abs(np.take_along_axis(imp - ref, top, 1)).max())
    print("%s: kept-query gap %.2e, GPU score error %.2e (margin %.0f); pair gap %.2e, GPU "
          "score error %.2e (margin %.0f)" % (name, float(fx["keep_gap"]), e_keep,
                                              float(fx["keep_gap"]) / max(e_keep, 1e-12),
                                              float(fx["pair_gap"]), e_pair,
                                              float(fx["pair_gap"]) / max(e_pair, 1e-12)))
    assert e_keep < float(fx["keep_gap"]) / 4 and e_pair < float(fx["pair_gap"]) / 4
    kept_tokens = np.take_along_axis(got_prop, pl.keep.cpu().numpy(), 1)
    assert np.array_equal(kept_tokens, np.take_along_axis(fx["proposals"], fx["keep_index"], 1))
    assert np.array_equal(pl.topk_idx.cpu().numpy(), fx["topk_idx"])
    assert np.array_equal(pl.sub_pos.cpu().numpy(), fx["sub_pos"])
    assert np.array_equal(pl.obj_pos.cpu().numpy(), fx["obj_pos"])
    errs = {k: _err(cls[k], fx["cls_" + k]) for k in ("sub", "obj", "cls", "rel", "importance")}
    errs.update({k: _err(box[k], fx["bbox_" + k]) for k in ("bbox", "sub_bbox", "obj_bbox")})
    errs["enc_cls"] = _err(cls["enc_cls_scores"].flatten()[torch.from_numpy(fx["enc_probe_idx"]).to(DEV)],
                           fx["enc_cls_probe"])
    errs["enc_box"] = _err(cls["enc_bbox_preds"].flatten()[torch.from_numpy(fx["box_probe_idx"]).to(DEV)],
                           fx["enc_box_probe"])
    print(name, "errors:", errs)
    assert all(e < 1e-3 for e in errs.values()), errs
    res = head.get_bboxes(cls, box, metas, rescale=True)
    for i, r in enumerate(res):
        assert len(r) == 6
        assert np.array_equal(r[1].cpu().numpy(), fx["res%d_labels" % i])
        assert np.array_equal(r[2].cpu().numpy(), fx["res%d_pairs" % i])
        assert _err(r[0], fx["res%d_det" % i]) < 2e-2          # pixels (and a score column)
        assert _err(r[5], fx["res%d_r_dists" % i]) < 1e-3


def test_bbox_head_graph_replay_and_other_layouts_are_bitwise_the_eager_result(built_lib):
    """hipGraph replay of the whole head, and feats handed over as plain NCHW tensors (copied
    into token rows) instead of the neck's in-place views, give the same bits."""
    fx, head, cls, box, metas = _run_fixture("bbox_small")
    keep = {k: v.clone() for k, v in list(cls.items()) + list(box.items())}
    H, W = int(fx["height"]), int(fx["width"])
    _, _, _, nsd, _ = _oracles(int(fx["weight_seed"]), int(fx["neck_seed"]))
    P, _, ncfg = _cfgs()
    neck = P.ChannelMapper(**ncfg).to(DEV)
    neck.load_state_dict(nsd)
    nf = neck([f.to(DEV) for f in _feats(fx["feat_seeds"], H, W)])
    plain = [f.contiguous() for f in nf]
    c2, b2 = head(plain, metas)
    assert head._last_plan.own_tokens
    for k, v in list(c2.items()) + list(b2.items()):
        assert torch.equal(v, keep[k]), k
    head.use_graphs = True
    for _ in range(3):
        c3, b3 = head(plain, metas)
    assert head._last_plan.graph_a is not None and head._last_plan.graph_b is not None
    for k, v in list(c3.items()) + list(b3.items()):
        assert torch.equal(v, keep[k]), k


def test_bbox_head_under_the_multi_stream_pipeline_is_bitwise_the_plain_call(built_lib):
    """`PipelinedHead` (pipeline.py) schedules CrossHeadBBox like CrossHead2: stage A = encoder
    + proposals, stage B = the query chain; neck + head of consecutive batches on alternating
    streams with per-slot buffers, the bench.py --head bbox loop."""
    import pairnet_amd as P
    fx, head, cls, box, metas = _run_fixture("bbox_small")
    want = head.get_bboxes(cls, box, metas, rescale=True)
    want = [[t.clone() for t in r] for r in want]
    H, W = int(fx["height"]), int(fx["width"])
    _, _, _, nsd, _ = _oracles(int(fx["weight_seed"]), int(fx["neck_seed"]))
    _, _, ncfg = _cfgs()
    neck = P.ChannelMapper(**ncfg).to(DEV)
    neck.load_state_dict(nsd)
    ins = [f.to(DEV) for f in _feats(fx["feat_seeds"], H, W)]
    noise = [torch.randn_like(f) for f in ins]
    head.use_graphs = True
    eng = P.PipelinedHead(head, depth=4, a_streams=2)
    outs = []
    for i in range(9):          # the fixture's input every third batch, noise otherwise
        sa = eng.streams_a[i % 2]
        with torch.cuda.stream(sa):
            r = eng.submit(neck(ins if i % 3 == 0 else noise, slot=i % 4), metas, rescale=True)
            if r is not None:
                outs.append([[t.clone() for t in x] for x in r])
    with torch.cuda.stream(eng.streams_a[0]):
        outs += [[[t.clone() for t in x] for x in r] for r in eng.flush()]
    torch.cuda.synchronize()
    assert len(outs) == 9
    for i in (0, 3, 6):
        for a, b in zip(outs[i], want):
            assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_proposals_tied_at_the_invalid_token_score_are_interchangeable(built_lib):
    """Tokens whose proposal box leaves (0.01, 0.99) are zeroed before `enc_output`, so they all
    carry ONE class score.  With these weights that score ranks inside the best 300, the
    selection cuts through the tie, and which tied tokens are taken is unspecified in the
    reference (torch.topk) -- but they yield identical queries (same embedding, same +inf box
    logits), so the multiset of decoder outputs must still be the reference's."""
    import pairnet_amd as P
    ohead, oneck, sd, nsd, _ = _oracles(31, 32)
    head, neck = _hip_models(sd, nsd)
    H, W = 608, 1023
    feats = seeded.smooth_feats(2608, 1, H, W, 8)[1:]
    metas = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=[1.3] * 4)]
    tr = {}
    with torch.no_grad():
        ohead(oneck(feats), metas, trace=tr)
    head(neck([f.to(DEV) for f in feats]), metas)
    pl = head._last_plan
    valid = P.CrossHeadBBox.proposals(pl.shapes)[1]
    n_inv = int((~valid[tr["topk_proposals"][0]]).sum())
    assert 0 < n_inv < 300 and int((~valid).sum()) > n_inv      # the cut goes through the tie
    assert int((~valid[pl.top_idx[0].cpu()]).sum()) == n_inv
    assert _err(pl.qscore.sort(-1)[0], tr["query_score"].sort(-1)[0]) < 1e-6
    assert _err(pl.ref[-1].view(1, -1, 4).sum(-1).sort(-1)[0], tr["coords"][-1].sum(-1).sort(-1)[0]) < 2e-5


def test_bbox_head_rejects_what_it_does_not_build(built_lib):
    P, cfg, _ = _cfgs()
    with pytest.raises(NotImplementedError):
        P.CrossHeadBBox(**dict(cfg, as_two_stage=False))
    with pytest.raises(ValueError):
        P.CrossHeadBBox(**dict(cfg, transformer=None))



def test_padded_batch_follows_the_key_padding_masks(built_lib):
    """Two images of different sizes in one batch (pairnet_bbox_head.py:196-213): masked sine
    encoding, valid-ratio reference points, zeroed value rows and proposals of padded tokens --
    the general path (`pn_token_sampling_f32` + `pn_msda_loc_f32`) against the oracle, whose
    padded-batch arithmetic is what tests/test_oracle.py pins to transformers' Deformable DETR
    and to the reference class."""
    ohead, oneck, sd, nsd, _ = _oracles(12347, 12348)
    head, neck = _hip_models(sd, nsd)
    H, W = 160, 192
    feats = _feats([100, 101], H, W)
    metas = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=[1.0] * 4),
             dict(batch_input_shape=(H, W), img_shape=(130, 150, 3), scale_factor=[1.5] * 4)]
    tr = {}
    with torch.no_grad():
        oc, ob = ohead(oneck(feats), metas, trace=tr)
        ores = ohead.get_bboxes(oc, ob, metas, rescale=True)
    hc, hb = head(neck([f.to(DEV) for f in feats]), metas)
    pl = head._last_plan
    assert pl.padded and tuple(pl.vr.shape) == (2, 4, 2) and float(pl.vr[0].min()) == 1.0
    # (padded tokens are masked out of every consumer; their own encodings are sin / cos of
    # -0.5 / eps in the reference: not compared)
    tv = pl.tok_valid.bool()
    assert _err(pl.X[tv], tr["memory"][tv.cpu()]) < 5e-5 and int((~tv).sum()) == 192
    assert _err(hc["enc_cls_scores"], oc["enc_cls_scores"]) < 5e-5
    fin = torch.isfinite(torch.logit(oc["enc_bbox_preds"])).all(-1)
    assert 0 < int((~fin[1]).sum()) and int((~fin[0]).sum()) == 0      # padded tokens: +inf boxes
    assert _err(hc["enc_bbox_preds"], oc["enc_bbox_preds"]) < 1e-5
    # proposals: the same VALID tokens and the same number of zeroed ones (those tie exactly
    # and are interchangeable, see the test above)
    got, want = pl.top_idx.cpu(), tr["topk_proposals"]
    ok = pl.valid.cpu().bool()
    for i in range(2):
        g, w_ = got[i], want[i]
        assert set(g[ok[i][g]].tolist()) == set(w_[ok[i][w_]].tolist())
        assert int((~ok[i][g]).sum()) == int((~ok[i][w_]).sum())
    assert _err(pl.qscore.sort(-1)[0], tr["query_score"].sort(-1)[0]) < 1e-6
    for k in ("sub", "obj", "cls", "rel", "importance"):
        assert _err(hc[k], oc[k]) < 1e-3, k
    for k in hb:
        assert _err(hb[k], ob[k]) < 1e-4, k
    # (pair ranking: unseparated weights here, near-ties allowed -- the golden fixtures above
    # assert exact indices; the selected SCORES must agree)
    top = lambda t: t.flatten(1).topk(100)[0]
    assert _err(top(hc["importance"]), top(oc["importance"])) < 1e-4
    res = head.get_bboxes(hc, hb, metas, rescale=True)
    assert all(len(r) == 6 and tuple(r[0].shape) == (200, 5) for r in res)
    # the same two images unpadded, one by one, give image 0's outputs again (padding of
    # the OTHER image does not leak)
    h0c, h0b = head(neck([f[:1].to(DEV) for f in feats]), metas[:1])
    assert _err(h0c["rel"], oc["rel"][:1]) < 1e-3 and not head._last_plan.padded


def test_detector_with_neck_and_bbox_head_end_to_end(built_lib):
    """build_detector(cross_r101_vg): native ResNet-101 (C3-C5) -> ChannelMapper -> CrossHeadBBox
    -> triplet2Result without masks (psgtr.py:53-71), against the oracles chained the same way."""
    import pairnet_amd as P
    from oracle.backbone import OracleResNet50, seeded_backbone_state
    det = P.build_detector(P.cross_r101_vg().model).to(DEV)
    ohead, oneck, sd, nsd, _ = _oracles(21, 22)
    det.bbox_head.load_state_dict(sd)
    det.neck.load_state_dict(nsd)
    obb = OracleResNet50(depth=101).eval()
    bsd = seeded_backbone_state(23, depth=101)
    obb.load_state_dict(bsd)
    det.backbone.load_state_dict(bsd)
    H, W = 224, 256
    img = seeded.uniform(np.random.default_rng(24), (1, 3, H, W), -2.0, 2.0)
    metas = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=[1.0] * 4)]
    out = det.simple_test(img.to(DEV), metas, rescale=True)
    assert len(out) == 1 and out[0].pan_results is None and out[0].masks is None
    assert out[0].refine_bboxes.shape == (200, 5) and out[0].rel_dists.shape == (100, 51)
    with torch.no_grad():
        feats = obb(img)[1:]
        cls, box = ohead(oneck(feats), metas)
    hc, hb = det.bbox_head._outputs(det.bbox_head._last_plan)
    assert _err(hc["enc_cls_scores"], cls["enc_cls_scores"]) < 1e-3
