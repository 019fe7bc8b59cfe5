"""The PSG evaluator's triplet matching (pairnet/evaluation/sgg_metrics.py:173-252,
:1276-1380).  CPU: oracle/evaluation.py against the reference's own functions (imported
from /root/reference under name-only stubs; build container only).  GPU: the device feed
(pairnet_amd.evaluation.TripletEvaluator) against the oracle on a crafted scene with exact
matches, near-threshold IoUs, wrong classes, duplicates and empty masks."""
import os
import sys
import types
import importlib.util

import numpy as np
import pytest
import torch

from oracle import evaluation as OE
from oracle import ref_shim


def _scene(seed, R=100, H=48, W=64, nobj=7, G=9):
    """Ground truth of rectangles + predictions derived from it with perturbations."""
    rng = np.random.default_rng(seed)
    gt_masks = np.zeros((nobj, H, W), bool)
    for i in range(nobj):
        y, x = rng.integers(0, H - 16), rng.integers(0, W - 16)
        gt_masks[i, y:y + rng.integers(6, 16), x:x + rng.integers(6, 16)] = True
    gt_labels = rng.integers(1, 134, nobj)
    pairs = [(s, o) for s in range(nobj) for o in range(nobj) if s != o]
    sel = rng.choice(len(pairs), G, replace=False)
    gt_rels = np.array([[pairs[j][0], pairs[j][1], rng.integers(1, 57)] for j in sel])
    labels = rng.integers(1, 134, 2 * R)
    masks = rng.random((2 * R, H, W)) > 0.97
    rel_dists = rng.random((R, 57)).astype(np.float32)
    rel_dists[:, 0] = 0
    for j, (s, o, pr) in enumerate(gt_rels):          # plant hits and near misses
        for rep, r in enumerate((3 * j, 3 * j + 1, 3 * j + 40)):
            labels[r], labels[R + r] = gt_labels[s], gt_labels[o]
            rel_dists[r, pr] = 2.0 if rep != 1 else 0.0          # rep 1: wrong predicate
            shift = (0, 1, 4)[rep]                               # rep 2: IoU near / below 0.5
            masks[r] = np.roll(gt_masks[s], shift, axis=1)
            masks[R + r] = np.roll(gt_masks[o], shift, axis=0)
    masks[7] = False                                             # an empty predicted mask
    rel_pairs = np.stack([np.arange(R), np.arange(R) + R], 1)
    return labels, rel_pairs, rel_dists, masks, gt_rels, gt_labels, gt_masks


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
def test_evaluation_oracle_equals_the_reference_functions():
    sys.dont_write_bytecode = True
    for name, attrs in (("mmdet", {}), ("mmdet.core", dict(bbox_overlaps=None)),
                        ("terminaltables", dict(AsciiTable=None))):
        m = sys.modules.setdefault(name, types.ModuleType(name))   # (ref_shim may have put
        for k, v in attrs.items():                                 # its own stubs there)
            if not hasattr(m, k):
                setattr(m, k, v)
    pkg = types.ModuleType("refeval")
    pkg.__path__ = [os.path.join(ref_shim.REF_ROOT, "pairnet/evaluation")]
    sys.modules["refeval"] = pkg
    mods = {}
    for n in ("sgg_eval_util", "sgg_metrics"):
        spec = importlib.util.spec_from_file_location(
            "refeval." + n, os.path.join(ref_shim.REF_ROOT, "pairnet/evaluation", n + ".py"))
        mods[n] = importlib.util.module_from_spec(spec)
        sys.modules["refeval." + n] = mods[n]
        spec.loader.exec_module(mods[n])
    M = mods["sgg_metrics"]
    for seed in (1, 2):
        labels, rel_pairs, rel_dists, masks, gt_rels, gt_labels, gt_masks = _scene(seed)
        ours = OE.evaluate(labels, rel_pairs, rel_dists, masks, gt_rels, gt_labels, gt_masks)
        pred_rels = np.column_stack((rel_pairs, 1 + rel_dists[:, 1:].argmax(1)))
        gt_t, gt_tm, _ = M._triplet_panseg(gt_rels, gt_labels, gt_masks)
        p_t, p_tm, _ = M._triplet_panseg(pred_rels, labels, masks)
        with np.errstate(invalid="ignore", divide="ignore"):
            for ph, key in ((False, "pred_to_gt"), (True, "phrdet_pred_to_gt")):
                ref = M._compute_pred_matches_panseg(gt_t, p_t, gt_tm, p_tm, 0.5, phrdet=ph)
                assert ref == ours[key]
        rec = M.SGRecall({}, {}, [], detection_method="pan_seg")
        rec.register_container("sgdet")
        rec._calculate_single(rec.result_dict, ours["pred_to_gt"], gt_rels, "sgdet")
        assert {k: v[0] for k, v in rec.result_dict["sgdet_recall"].items()} == ours["sgdet_recall"]
        assert sum(len(x) for x in ours["pred_to_gt"]) >= 9      # the scene really has matches
        # the subject / object IoU statistic (:1087-1131)
        with np.errstate(invalid="ignore", divide="ignore"):
            rs, ro = M._compute_iou_panseg(gt_t, labels, gt_tm, masks)
            os_, oo = OE.iou_panseg(gt_t, labels, gt_tm, masks)
        assert np.array_equal(rs, os_) and np.array_equal(ro, oo) and len(rs) > 0
    # ---- dataset level: SGMeanRecall (:669-916) over several images ----
    num_rel = 57
    mr = M.SGMeanRecall({}, {}, [], num_rel, ["bg"] + ["p%d" % i for i in range(1, num_rel)],
                        detection_method="pan_seg")
    mr.register_container("sgdet")
    collected, recalls = [], {k: [] for k in (20, 50, 100)}
    from pairnet_amd.evaluation import SceneGraphMetrics
    agg = SceneGraphMetrics(num_predicates=56)
    for seed in (1, 2, 3, 4):
        labels, rel_pairs, rel_dists, masks, gt_rels, gt_labels, gt_masks = _scene(seed)
        ours = OE.evaluate(labels, rel_pairs, rel_dists, masks, gt_rels, gt_labels, gt_masks)
        local = dict(pred_to_gt=ours["pred_to_gt"], phrdet_pred_to_gt=ours["phrdet_pred_to_gt"],
                     gt_rels=gt_rels)
        mr.collect_mean_recall_items({}, local, "sgdet")
        collected.append((OE.mean_recall_collect(ours["pred_to_gt"], gt_rels, num_rel),
                          OE.mean_recall_collect(ours["phrdet_pred_to_gt"], gt_rels, num_rel)))
        agg.add(ours, gt_rels)
        holder = types.SimpleNamespace()
        M.SGPairAccuracy.prepare_gtpair(holder, dict(pred_rel_inds=rel_pairs, gt_rels=gt_rels))
        assert np.array_equal(holder.pred_pair_in_gt, OE.pred_pair_in_gt(rel_pairs, gt_rels))
        assert np.array_equal(holder.pred_pair_in_gt,
                              SceneGraphMetrics.pred_pair_in_gt(rel_pairs, gt_rels))
    agg.add(dict(pred_to_gt=[], phrdet_pred_to_gt=[], sgdet_recall=None, phrdet_recall=None),
            np.zeros((0, 3), int))                              # an image without relations
    mr.calculate_mean_recall("sgdet")
    for mode, j in (("sgdet", 0), ("phrdet", 1)):
        want, want_list = mr.result_dict[mode + "_mean_recall"], mr.result_dict[mode + "_mean_recall_list"]
        got, got_list = OE.mean_recall([c[j] for c in collected], num_rel)
        assert got == want and got_list == want_list
        summ = agg.summary()
        assert summ[mode + "_mean_recall"] == want and summ[mode + "_mean_recall_list"] == want_list
    assert summ["images"] == 4 and summ["skipped"] == 1 and summ["sgdet_mean_recall"][100] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_device_evaluator_feed_equals_the_oracle(seed):
    from pairnet_amd.evaluation import TripletEvaluator
    labels, rel_pairs, rel_dists, masks, gt_rels, gt_labels, gt_masks = _scene(seed, H=61, W=83)
    ref = OE.evaluate(labels, rel_pairs, rel_dists, masks, gt_rels, gt_labels, gt_masks)
    dev = "cuda:0"
    result = (None, torch.from_numpy(labels).to(dev), torch.from_numpy(rel_pairs),
              torch.from_numpy(masks).to(dev), None, None, None,
              torch.from_numpy(rel_dists).to(dev))
    out = TripletEvaluator()(result, gt_rels, gt_labels, gt_masks)
    assert out["pred_to_gt"] == ref["pred_to_gt"]
    assert out["phrdet_pred_to_gt"] == ref["phrdet_pred_to_gt"]
    assert out["sgdet_recall"] == ref["sgdet_recall"]
    assert out["phrdet_recall"] == ref["phrdet_recall"]
    assert out["sgdet_recall"][100] > 0.5                        # (the planted hits are found)
    # the IoU statistic from the same device popcounts
    ev = TripletEvaluator()
    gt_t, gt_tm = OE.triplets(gt_rels, gt_labels, gt_masks)
    with np.errstate(invalid="ignore", divide="ignore"):
        want_s, want_o = OE.iou_panseg(gt_t, labels, gt_tm, masks)
    got_s, got_o = ev.iou_stats(result, gt_rels, gt_labels, gt_masks)
    assert np.array_equal(got_s, want_s) and np.array_equal(got_o, want_o)
    # an image without ground-truth relations is skipped, not an error
    empty = ev(result, np.zeros((0, 3), int), gt_labels, gt_masks)
    assert empty["sgdet_recall"] is None and len(empty["pred_to_gt"]) == 100


# ---- detection_method == "bbox": the box-trunk sibling head's results ----------------------
def _box_scene(seed, R=100, nobj=8, G=10, W=640.0, H=480.0):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0, [W - 120, H - 120], (nobj, 2))
    wh = rng.uniform(20, 120, (nobj, 2))
    gt_boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    gt_labels = rng.integers(1, 151, nobj)
    pairs = [(s, o) for s in range(nobj) for o in range(nobj) if s != o]
    sel = rng.choice(len(pairs), G, replace=False)
    gt_rels = np.array([[pairs[j][0], pairs[j][1], rng.integers(1, 51)] for j in sel])
    labels = rng.integers(1, 151, 2 * R)
    xy = rng.uniform(0, [W - 60, H - 60], (2 * R, 2))
    boxes = np.concatenate([xy, xy + rng.uniform(5, 60, (2 * R, 2))], 1).astype(np.float32)
    boxes[5] = [10, 10, 10, 10]                                   # a degenerate box (zero area)
    rel_dists = rng.random((R, 51)).astype(np.float32)
    rel_dists[:, 0] = 0
    for j, (s, o, pr) in enumerate(gt_rels):          # hits, wrong predicate, IoU around 0.5
        for rep, r in enumerate((3 * j, 3 * j + 1, 3 * j + 2, 3 * j + 40)):
            labels[r], labels[R + r] = gt_labels[s], gt_labels[o]
            rel_dists[r, pr] = 2.0 if rep != 1 else 0.0
            # shrink so that IoU = f: rep 2 sits exactly ON the threshold region (w * 0.5)
            f = (1.0, 1.0, 0.5, 0.45)[rep]
            for src, dst in ((s, r), (o, R + r)):
                b = gt_boxes[src].copy()
                b[2] = b[0] + (b[2] - b[0]) * f
                boxes[dst] = b
    scores = rng.random(2 * R).astype(np.float32)
    det = np.concatenate([boxes, scores[:, None]], 1)
    rel_pairs = np.stack([np.arange(R), np.arange(R) + R], 1)
    return labels, rel_pairs, rel_dists, det, gt_rels, gt_labels, gt_boxes


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
def test_box_evaluation_oracle_equals_the_reference_functions():
    """oracle.evaluation.pred_matches_bbox against the reference's `_compute_pred_matches_bbox`
    / `_triplet_bbox` (sgg_metrics.py:1181-1273) run from /root/reference, with mmdet's
    `bbox_overlaps` (absent) supplied by the restatement the oracle itself uses."""
    sys.dont_write_bytecode = True
    for name, attrs in (("mmdet", {}), ("mmdet.core", dict(bbox_overlaps=OE.bbox_overlaps)),
                        ("terminaltables", dict(AsciiTable=None))):
        m = sys.modules.setdefault(name, types.ModuleType(name))
        for k, v in attrs.items():
            setattr(m, k, v)
    pkg = types.ModuleType("refevalb")
    pkg.__path__ = [os.path.join(ref_shim.REF_ROOT, "pairnet/evaluation")]
    sys.modules["refevalb"] = pkg
    mods = {}
    for n in ("sgg_eval_util", "sgg_metrics"):
        spec = importlib.util.spec_from_file_location(
            "refevalb." + n, os.path.join(ref_shim.REF_ROOT, "pairnet/evaluation", n + ".py"))
        mods[n] = importlib.util.module_from_spec(spec)
        sys.modules["refevalb." + n] = mods[n]
        spec.loader.exec_module(mods[n])
    M = mods["sgg_metrics"]
    for seed in (1, 2):
        labels, rel_pairs, rel_dists, det, gt_rels, gt_labels, gt_boxes = _box_scene(seed)
        ours = OE.evaluate_boxes(labels, rel_pairs, rel_dists, det[:, :4], gt_rels, gt_labels, gt_boxes)
        pred_rels = np.column_stack((rel_pairs, 1 + rel_dists[:, 1:].argmax(1)))
        gt_t, gt_tb, _ = M._triplet_bbox(gt_rels, gt_labels, gt_boxes)
        p_t, p_tb, _ = M._triplet_bbox(pred_rels, labels, det[:, :4])
        for ph, key in ((False, "pred_to_gt"), (True, "phrdet_pred_to_gt")):
            ref = M._compute_pred_matches_bbox(gt_t, p_t, gt_tb, p_tb, 0.5, phrdet=ph)
            assert ref == ours[key]
        assert sum(len(x) for x in ours["pred_to_gt"]) >= 10


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_device_box_evaluator_feed_equals_the_oracle(seed):
    from pairnet_amd.evaluation import TripletEvaluator
    labels, rel_pairs, rel_dists, det, gt_rels, gt_labels, gt_boxes = _box_scene(seed)
    ref = OE.evaluate_boxes(labels, rel_pairs, rel_dists, det[:, :4], gt_rels, gt_labels, gt_boxes)
    dev = "cuda:0"
    result = (torch.from_numpy(det).to(dev), torch.from_numpy(labels).to(dev),
              torch.from_numpy(rel_pairs), None, None, torch.from_numpy(rel_dists).to(dev))
    out = TripletEvaluator().evaluate_boxes(result, gt_rels, gt_labels, gt_boxes)
    for k in ("pred_to_gt", "phrdet_pred_to_gt", "sgdet_recall", "phrdet_recall"):
        assert out[k] == ref[k], k
    assert out["sgdet_recall"][100] > 0.5
    ig = TripletEvaluator().pred_to_gt(TripletEvaluator().match_boxes(
        result, gt_rels, gt_labels, gt_boxes, ignore_rel=True))
    trip = lambda rel, cls, bx: (np.column_stack((cls[rel[:, 0]], rel[:, 2], cls[rel[:, 1]])),
                                 np.column_stack((bx[rel[:, 0]], bx[rel[:, 1]])))
    pred_rels = np.column_stack((rel_pairs, 1 + rel_dists[:, 1:].argmax(1)))
    gt_t, gt_tb = trip(gt_rels, gt_labels, gt_boxes)
    p_t, p_tb = trip(pred_rels, labels, det[:, :4])
    assert ig == OE.pred_matches_bbox(gt_t, p_t, gt_tb, p_tb, 0.5, ignore_rel=True)
