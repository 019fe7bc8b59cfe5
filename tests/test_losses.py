"""CPU: the training-side loss FORWARD of CrossHead2 (first slice).

`oracle/losses.py` (our restatement of pairnet_head.py:419-718, IdMatcher, BCEWithLogitsLoss)
against the reference's OWN methods executed from /root/reference under
oracle/ref_shim.install_training() -- same inputs, same torch RNG state for the sampled mask
points -- bit for bit; and the [3P] matching costs / point sampling of oracle/mmdet_train.py
against HuggingFace `transformers`' independent Mask2Former loss utilities."""
import numpy as np
import pytest
import torch

from oracle import mmdet_train as T
from oracle import ref_shim
from oracle.losses import OracleCrossHead2Loss

needs_ref = pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")


def _case(seed, H=96, W=128, bs=2):
    g = torch.Generator().manual_seed(seed)
    feats = [torch.randn(bs, c, H // s, W // s, generator=g)
             for c, s in zip((256, 512, 1024, 2048), (4, 8, 16, 32))]
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0] * 4)] * bs
    gt_labels = [torch.tensor([3, 17, 90, 120, 3]), torch.tensor([5, 60])][:bs]
    gt_masks = [torch.rand(5, H, W, generator=g) > 0.6, torch.rand(2, H, W, generator=g) > 0.5][:bs]
    gt_rels = [torch.tensor([[0, 1, 5], [2, 3, 17], [1, 0, 56], [4, 2, 5], [0, 1, 9]]),
               torch.tensor([[0, 1, 2]])][:bs]
    return feats, metas, gt_rels, gt_labels, gt_masks


@needs_ref
@pytest.mark.parametrize("seed", [1, 2])
def test_oracle_loss_equals_the_reference_methods_bit_for_bit(seed):
    torch.manual_seed(0)
    head = ref_shim.build_reference_training_head()
    feats, metas, gt_rels, gt_labels, gt_masks = _case(seed)
    with torch.no_grad():
        cls, masks = head.forward(feats, metas)
        torch.manual_seed(100 + seed)
        want = head.loss(cls, masks, gt_rels, None, gt_labels, gt_masks, metas)
        mine = OracleCrossHead2Loss.from_config(ref_shim.reference_model_cfg())
        torch.manual_seed(100 + seed)
        trace = {}
        got = mine.loss(cls, masks, gt_rels, gt_labels, gt_masks, trace=trace)
    assert set(got) == set(want) == {"loss_r_cls", "loss_sub_cls", "loss_obj_cls", "loss_match"}
    for k in want:
        assert torch.equal(got[k], want[k]), (k, float(got[k]), float(want[k]))
        assert torch.isfinite(got[k]) and float(got[k]) > 0
    # the targets behind them: every ground-truth triplet found a relation query, the
    # importance matrix holds one entry per distinct (subject query, object query) pair
    assert int((torch.cat(trace["weights"]) > 0).sum()) == sum(len(r) for r in gt_rels)
    assert float(trace["importance"].sum()) <= sum(len(r) for r in gt_rels)
    # Seesaw's persistent label counts moved identically in both objects
    assert torch.equal(mine.rel_loss.cum_samples, head.rel_cls_loss.cum_samples)
    # explicit sample points reproduce the run without touching the RNG
    pts = [im["point_coords"] for im in trace["images"]]
    mine2 = OracleCrossHead2Loss.from_config(ref_shim.reference_model_cfg())
    again = mine2.loss(cls, masks, gt_rels, gt_labels, gt_masks, point_coords=pts)
    for k in want:
        assert torch.equal(again[k], want[k])


@needs_ref
def test_image_without_relations_and_reference_id_matcher():
    """`IdMatcher.assign` of the reference (approaches/matcher.py:208-275) against the
    restated matching, including the empty ground truth (every query background)."""
    import sys
    ref_shim.install_training()
    IdMatcher = sys.modules["pairnet.models.relation_heads.approaches.matcher"].IdMatcher
    ref = IdMatcher(sub_id_cost=dict(type="ClassificationCost", weight=1.0),
                    obj_id_cost=dict(type="ClassificationCost", weight=1.0),
                    r_cls_cost=dict(type="ClassificationCost", weight=0.0))
    mine = OracleCrossHead2Loss()
    g = torch.Generator().manual_seed(5)
    sub, obj, rel = (torch.randn(100, n, generator=g) for n in (134, 134, 56))
    for G in (0, 1, 7):
        gs, go = torch.randint(0, 133, (G,), generator=g), torch.randint(0, 133, (G,), generator=g)
        gr = torch.randint(0, 56, (G,), generator=g)
        want = ref.assign(sub, obj, rel, gs, go, gr, None)
        if G == 0:
            # reference quirk: the empty case returns a PAIR of results (a leftover of the old
            # matcher, matcher.py:246-248), which `_get_target_single` hands to the sampler as
            # is -- the reference's loss fails with an AttributeError on an image without
            # ground-truth relations (training sets drop those images)
            assert isinstance(want, tuple) and torch.equal(want[0].gt_inds, want[1].gt_inds)
            want = want[0]
        got = mine.id_match(sub, obj, rel, gs, go, gr)
        assert torch.equal(got, want.gt_inds) and int((got > 0).sum()) == G
    with pytest.raises(ValueError):        # ours says why, instead of the AttributeError
        mine.targets_single(sub, obj, torch.randn(100, 134, generator=g),
                            torch.randn(100, 8, 8, generator=g), rel, torch.zeros(0, 3),
                            torch.tensor([3]), torch.ones(1, 8, 8) > 0)


def test_matching_costs_and_point_sampling_against_huggingface_mask2former():
    """[3P] pin: mmdet's CrossEntropyLossCost / DiceCost and mmcv's point_sample (restated in
    oracle/mmdet_train.py) against the same quantities of transformers' Mask2Former loss."""
    hf = pytest.importorskip("transformers.models.mask2former.modeling_mask2former")
    g = torch.Generator().manual_seed(7)
    pred = torch.randn(100, 12544, generator=g) * 3
    tgt = (torch.rand(6, 12544, generator=g) > 0.7).float()
    ce = T.CrossEntropyLossCost(weight=1.0, use_sigmoid=True)(pred, tgt)
    assert torch.allclose(ce, hf.pair_wise_sigmoid_cross_entropy_loss(pred, tgt), rtol=1e-5, atol=1e-6)
    dice = T.DiceCost(weight=1.0, pred_act=True, eps=1.0)(pred, tgt)
    assert torch.allclose(dice, hf.pair_wise_dice_loss(pred, tgt), rtol=1e-5, atol=1e-6)
    maps = torch.randn(5, 1, 24, 32, generator=g)
    pts = torch.rand(5, 300, 2, generator=g)
    assert torch.equal(T.point_sample(maps, pts), hf.sample_point(maps, pts, align_corners=False))


def test_mmdet_cross_entropy_mean_is_over_elements():
    """mmdet's CrossEntropyLoss multiplies element losses by the class weight and divides by
    the number of ELEMENTS (torch divides by the summed weights): 0.1-weighted background."""
    loss = T.CrossEntropyLoss(class_weight=[1.0, 1.0, 0.1], loss_weight=2.0)
    x = torch.tensor([[2.0, 0.5, 0.1], [0.2, 0.1, 3.0]])
    y = torch.tensor([0, 2])
    el = torch.nn.functional.cross_entropy(x, y, reduction="none") * torch.tensor([1.0, 0.1])
    assert torch.allclose(loss(x, y), 2.0 * el.mean())


def test_seesaw_loss_properties():
    """Seesaw (restated, unpinned): with empty history and q = 0 it is plain cross entropy;
    accumulated counts of a frequent class lower the penalty on rarer classes' logits."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(8, 56 + 2, generator=g)
    y = torch.tensor([1, 1, 1, 1, 1, 1, 4, 9])
    plain = T.SeesawLoss(num_classes=56, p=0.8, q=0.0, loss_weight=1.0)
    first = plain(x, y)["loss_cls_classes"]
    # (the first call already counts its own labels: class 1 has 6 samples, 4 and 9 one each)
    ce = torch.nn.functional.cross_entropy(x[:, :56], y)
    assert float(first) < float(ce)            # mitigation only ever shrinks negative logits
    assert plain.cum_samples[1] == 6 and plain.cum_samples[4] == 1
    full = T.SeesawLoss(num_classes=56, loss_weight=2.0)
    a, b = full(x, y)["loss_cls_classes"], full(x, y)["loss_cls_classes"]
    assert torch.isfinite(a) and torch.isfinite(b) and full.cum_samples[1] == 12


@pytest.mark.parametrize("h,w,H,W", [(90, 120, 96, 128), (96, 128, 96, 128), (50, 77, 96, 160),
                                      (61, 45, 75, 51)])
def test_ground_truth_mask_preparation_is_pad_then_nearest(h, w, H, W):
    """oracle.losses.prepare_gt_masks against the two torch calls PSGTr.forward_train makes
    (psgtr.py:132-138: F.pad to the batch tensor's size, F.interpolate(mode="nearest") to
    half of it), including odd batch sizes where the nearest source index is not 2 * dst."""
    import torch.nn.functional as F
    from oracle.losses import prepare_gt_masks
    g = torch.Generator().manual_seed(h * 1000 + w)
    mask = (torch.rand(4, h, w, generator=g) > 0.5).to(torch.uint8)
    want = F.interpolate(F.pad(mask, (0, W - w, 0, H - h)).unsqueeze(1), size=(H // 2, W // 2),
                         mode="nearest").squeeze(1)
    got = prepare_gt_masks(mask.numpy(), H, W)
    assert got.shape == tuple(want.shape) and got.dtype == np.uint8
    assert np.array_equal(got, want.numpy())
