"""GPU: the sibling head `CrossHeadBaseline` (reference relation_heads/baseline.py) on the
shared trunk, against the golden vectors recorded from the reference class and against
the CPU oracle; plus the kernels only this head uses (general top-k, softmax+foreground
pack, row argmax, triplet finish).

Tolerances: logits / scores within 1e-3 (fp32, north_star); argmax / top-k indices
bit-exact wherever the reference's own scores separate the candidates by more than
TIE_TOL (fixture match_gap ~1e-6: below fp32 re-association noise, so near-ties are
compared as sets)."""
import numpy as np
import pytest
import torch

from helpers import baseline_cfg, golden, oracle_baseline_head, overrides_of
from oracle import seeded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TIE_TOL = 2e-5


def _hip_head(sd):
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from pairnet_amd import CrossHeadBaseline
    head = CrossHeadBaseline(**baseline_cfg())
    head.load_state_dict(sd)
    return head.to(DEV)


def _err(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


# ------------------------------------------------------------------ kernels
@pytest.mark.parametrize("B,n,div,k", [(1, 5600, 56, 100), (3, 777, 7, 20), (2, 65536, 256, 256),
                                       (1, 5, 2, 5)])
def test_topk_f32_matches_torch(B, n, div, k):
    from pairnet_amd import hip
    g = torch.Generator().manual_seed(n + k)
    x = torch.randn(B, n, generator=g)
    tie = torch.arange(0, n - 1, 7)
    x[:, tie] = x[:, tie + 1]                                # exact ties: smaller index first
    xd = x.to(DEV)
    idx = torch.empty(B, k, dtype=torch.int64, device=DEV)
    quot, rem = torch.empty_like(idx), torch.empty_like(idx)
    hip.topk(xd, idx, quot, rem, B, n, div, k)
    order = torch.sort(x, dim=-1, descending=True, stable=True)[1][:, :k]
    assert torch.equal(idx.cpu(), order)
    assert torch.equal(quot.cpu(), order // div) and torch.equal(rem.cpu(), order % div)


def test_topk_f32_refuses_bad_sizes():
    from pairnet_amd import hip
    x = torch.zeros(4, device=DEV)
    i = torch.zeros(8, dtype=torch.int64, device=DEV)
    for n, div, k in ((4, 2, 5), (4, 0, 2), (70000, 56, 10), (4, 2, 0)):
        with pytest.raises(RuntimeError):
            hip.topk(x, i, i, i, 1, n, div, k)


def test_softmax_fg_argmax_and_triplet_finish():
    from pairnet_amd import hip
    g = torch.Generator().manual_seed(3)
    R, C, Q, k = 100, 57, 100, 100
    logits = torch.randn(R, C, generator=g) * 3
    probs = torch.empty(R, C, device=DEV)
    fg = torch.empty(R * (C - 1), device=DEV)
    hip.softmax_fg(logits.to(DEV), probs, fg, R, C)
    ref = torch.softmax(logits, -1)
    assert _err(probs, ref) < 1e-6
    assert torch.equal(fg.view(R, C - 1), probs[:, 1:])
    sc = torch.randn(R, Q, generator=g)
    sc[:, 40] = sc[:, 10]                                     # ties -> first index
    ids = torch.empty(R, dtype=torch.int64, device=DEV)
    hip.row_argmax(sc.to(DEV), ids, R, Q)
    want = torch.tensor([int(np.flatnonzero(r == r.max())[0]) for r in sc.numpy()])
    assert torch.equal(ids.cpu(), want)
    s_lab = torch.randint(0, 133, (R,), generator=g)
    o_lab = torch.randint(0, 133, (R,), generator=g)
    flat = torch.randperm(R * (C - 1), generator=g)[:k]
    tri, rem = flat // (C - 1), flat % (C - 1)
    labels = torch.empty(2 * k, dtype=torch.int64, device=DEV)
    r_labels = torch.empty(k, dtype=torch.int64, device=DEV)
    r_scores, r_dists = torch.empty(k, device=DEV), torch.empty(k, C, device=DEV)
    hip.triplet_finish(s_lab.to(DEV), o_lab.to(DEV), probs, tri.to(DEV), rem.to(DEV), labels,
                       r_labels, r_scores, r_dists, k, C)
    pc = probs.cpu()
    assert torch.equal(labels.cpu(), torch.cat((s_lab[tri] + 1, o_lab[tri] + 1)))
    assert torch.equal(r_labels.cpu(), rem + 1)
    assert torch.equal(r_dists.cpu(), pc[tri])
    assert torch.equal(r_scores.cpu(), pc[tri, rem + 1])


# ------------------------------------------------------------------ whole head
def _setup():
    fx = golden("baseline_small")
    head_o, sd, crc = oracle_baseline_head(int(fx["weight_seed"]), overrides_of(fx))
    assert crc == int(fx["weight_crc"])
    H, W, bs = int(fx["height"]), int(fx["width"]), int(fx["batch"])
    feats = seeded.seeded_feats(int(fx["feat_seed"]), bs, H, W)
    assert seeded.checksum(feats) == int(fx["feat_crc"])
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0, 2.0, 2.0, 2.0])] * bs
    return fx, head_o, sd, feats, metas


def _match_ids(ref_scores, got_ids, tol):
    """Every chosen id scores within tol of the reference's row maximum."""
    s = torch.as_tensor(ref_scores)
    chosen = torch.gather(s, -1, torch.as_tensor(got_ids).unsqueeze(-1)).squeeze(-1)
    return bool((chosen >= s.max(-1)[0] - tol).all())


@pytest.mark.parametrize("all_layers", [False, True])
def test_baseline_forward_against_reference_golden(all_layers):
    fx, head_o, sd, feats, metas = _setup()
    head = _hip_head(sd)
    head.return_all_layers = all_layers
    cls, masks = head.forward([f.to(DEV) for f in feats], metas)
    torch.cuda.synchronize()
    pl = head._last_plan
    bs = len(metas)
    assert set(cls) == {"sub", "obj", "cls", "rel", "subject_scores", "object_scores"}
    assert set(masks) == {"mask", "sub_seg", "obj_seg"}
    nl = 9 if all_layers else 1
    assert cls["cls"].shape == (nl, bs, 100, 134) and masks["mask"].shape[:3] == (nl, bs, 100)
    assert cls["rel"].shape == (bs, 100, 57)
    scale = max(1.0, float(np.abs(fx["mask_last"]).max()))
    errs = dict(rel=_err(cls["rel"], fx["cls_rel"]),
                cls_last=_err(cls["cls"][-1], fx["cls_cls"][-1]),
                sub_scores=_err(cls["subject_scores"], fx["cls_subject_scores"]),
                obj_scores=_err(cls["object_scores"], fx["cls_object_scores"]),
                mask_last=_err(masks["mask"][-1], fx["mask_last"]) / scale)
    if all_layers:
        errs["cls_all"] = _err(cls["cls"], fx["cls_cls"])
        probe = masks["mask"].flatten(1)[:, torch.from_numpy(fx["mask_probe_idx"]).to(DEV)]
        errs["mask_all"] = _err(probe, fx["mask_probe"]) / scale
    print("baseline_small errors:", errs)
    assert all(v < 1e-3 for v in errs.values()), errs
    # argmax matching: exact where the reference separates the candidates
    for name, ids in (("subject_scores", pl.sub_ids), ("object_scores", pl.obj_ids)):
        ref_ids = fx["sub_ids" if name[0] == "s" else "obj_ids"]
        same = (ids.cpu().numpy() == ref_ids).mean()
        print(name, "argmax identical: %.3f" % same)
        assert _match_ids(fx["cls_" + name], ids.cpu(), TIE_TOL)
        if float(fx["match_gap"]) > TIE_TOL:
            assert same == 1.0
    # the gathered outputs are the GPU's own rows
    sub = pl.sub_ids.cpu()
    assert torch.equal(cls["sub"].cpu(), torch.gather(
        cls["cls"][-1].cpu(), 1, sub[..., None].expand(-1, -1, 134)))
    obj = pl.obj_ids.cpu()
    hw = masks["mask"].shape[-2:]
    assert torch.equal(masks["obj_seg"].cpu(), torch.gather(
        masks["mask"][-1].cpu(), 1, obj[..., None, None].expand(-1, -1, hw[0], hw[1])))


def test_baseline_get_bboxes_on_reference_outputs():
    """Post-processing in isolation: the reference's own forward outputs go through the
    device get_bboxes; compared with the reference's result tuple."""
    fx, head_o, sd, feats, metas = _setup()
    head = _hip_head(sd)
    bs = len(metas)
    m_last = torch.from_numpy(fx["mask_last"])
    hw = m_last.shape[-2:]
    g = lambda ids: torch.gather(m_last, 1, torch.from_numpy(ids)[..., None, None].expand(
        -1, -1, hw[0], hw[1]))
    cls = {k: torch.from_numpy(fx["cls_" + k]).to(DEV) for k in ("sub", "obj", "rel")}
    cls["cls"] = torch.from_numpy(fx["cls_cls"][-1:]).to(DEV)
    masks = dict(mask=m_last.unsqueeze(0).to(DEV), sub_seg=g(fx["sub_ids"]).to(DEV),
                 obj_seg=g(fx["obj_ids"]).to(DEV))
    res = head.get_bboxes(cls, masks, metas)
    torch.cuda.synchronize()
    for i, r in enumerate(res):
        got_lab, ref_lab = r[6].cpu().numpy(), fx["res%d_r_labels" % i]
        assert _err(r[5], fx["res%d_r_scores" % i]) < 1e-6
        same = got_lab == ref_lab
        if float(fx["rank_gap"][i]) > 1e-6:       # GPU/CPU expf differ by ~1e-7
            assert same.all()
        both = np.concatenate([same, same])
        assert np.array_equal(r[1].cpu().numpy()[both], fx["res%d_labels" % i][both])
        assert _err(torch.from_numpy(r[7].cpu().numpy()[same]),
                    fx["res%d_r_dists" % i][same]) < 1e-6
        shape = tuple(fx["res%d_masks_shape" % i])
        ref_masks = np.unpackbits(fx["res%d_masks" % i])[:int(np.prod(shape))].reshape(shape)
        got = r[3].cpu().numpy()
        assert got.shape == shape and got.dtype == np.bool_
        assert (got[both] != ref_masks[both].astype(bool)).mean() < 1e-4
        assert (r[4].cpu().numpy() != fx["res%d_pan_img" % i]).mean() < 1e-3
        assert np.array_equal(r[2].numpy(), fx["res%d_rel_pairs" % i])
        assert r[0].shape == (200, 5)


@pytest.mark.parametrize("H,W", [(72, 104), (800, 1333)])
def test_baseline_simple_test_and_oracle_other_seed(H, W):
    """A second weight / input seed, batch 1, an odd size and the production shape: GPU vs the
    CPU oracle."""
    head_o, sd, _ = oracle_baseline_head(91)
    feats = seeded.seeded_feats(92, 1, H, W)
    metas = [dict(img_shape=(H, W, 3), scale_factor=[1.0] * 4)]
    head = _hip_head(sd)
    with torch.no_grad():
        cls_o, masks_o = head_o.forward(feats, metas)
    res = head.simple_test_bboxes([f.to(DEV) for f in feats], metas)
    torch.cuda.synchronize()
    cls, masks = head._outputs(head._last_plan)
    print("baseline %dx%d: rel %.2e cls %.2e scores %.2e" % (
        H, W, _err(cls["rel"], cls_o["rel"]), _err(cls["cls"][-1], cls_o["cls"][-1]),
        _err(cls["subject_scores"], cls_o["subject_scores"])))
    assert _err(cls["rel"], cls_o["rel"]) < 1e-3
    assert _err(cls["cls"][-1], cls_o["cls"][-1]) < 1e-3
    assert _err(cls["subject_scores"], cls_o["subject_scores"]) < 1e-3
    scale = max(1.0, float(masks_o["mask"][-1].abs().max()))
    assert _err(masks["mask"][-1], masks_o["mask"][-1]) < 1e-3 * scale
    assert _match_ids(cls_o["subject_scores"], head._last_plan.sub_ids.cpu(), TIE_TOL)
    r = res[0]
    h0, w0 = (H, W) if (H, W) == (72, 104) else r[4].shape
    assert r[1].shape == (200,) and r[3].shape == (200, h0, w0) and r[4].shape == (h0, w0)
    assert r[5].shape == (100,) and r[6].shape == (100,) and r[7].shape == (100, 57)
    s = r[5].cpu().numpy()
    assert (np.diff(s) <= 0).all()                       # ranked
    assert (r[6].cpu().numpy() >= 1).all() and (r[6].cpu().numpy() <= 56).all()
    # ranking vs the oracle's scores, tie-aware
    fg = torch.softmax(cls_o["rel"][0], -1)[:, 1:].reshape(-1)
    assert _err(r[5], fg.topk(100)[0]) < 1e-3


def test_baseline_graphs_and_pipeline_are_bitwise_the_eager_result():
    from pairnet_amd import PipelinedHead
    fx, head_o, sd, feats, metas = _setup()
    feats_d = [f.to(DEV) for f in feats]
    head = _hip_head(sd)
    ref = head.simple_test_bboxes(feats_d, metas)
    ref = [[t.clone() if t.is_cuda else t for t in r] for r in ref]
    torch.cuda.synchronize()
    head.use_graphs = True
    pipe = PipelinedHead(head, depth=3)
    outs = []
    for _ in range(5):
        o = pipe.submit(feats_d, metas)
        if o is not None:
            outs.append([[t.clone() if t.is_cuda else t for t in r] for r in o])
    for o in pipe.flush():
        outs.append(o)
    torch.cuda.synchronize()
    assert len(outs) == 5
    for o in outs:
        for ra, rb in zip(ref, o):
            for x, y in zip(ra, rb):
                assert torch.equal(x.cpu(), y.cpu())
