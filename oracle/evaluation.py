"""CPU restatement of the PSG evaluator's triplet matching (TEST INFRASTRUCTURE).

Follows pairnet/evaluation/sgg_metrics.py: `calculate_recall` :173-252 (predicted
relations :207-209), `_triplet_panseg` :1276-1308, `_compute_pred_matches_panseg`
:1311-1371, `mask_iou` :1374-1380, recall@K `_calculate_single` :95-99, and
sgg_eval_util.py `intersect_2d` :12-26.  Pinned against the reference functions
themselves, imported from /root/reference under name-only stubs of `mmdet.core` and
`terminaltables` (tests/test_evaluation.py, build container only).
"""
from functools import reduce

import numpy as np


def mask_iou(m1, m2):
    a1, a2 = np.count_nonzero(m1), np.count_nonzero(m2)
    inter = np.count_nonzero(np.logical_and(m1, m2))
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.float64(inter) / np.float64(a1 + a2 - inter)


def triplets(relations, classes, masks):
    sub, ob, pred = relations[:, 0], relations[:, 1], relations[:, 2]
    return (np.column_stack((classes[sub], pred, classes[ob])),
            np.stack((masks[sub], masks[ob]), axis=1))


def pred_matches(gt_trip, pred_trip, gt_masks, pred_masks, thr, phrdet=False, ignore_rel=False):
    if ignore_rel:
        gt_trip, pred_trip = gt_trip[:, [0, 2]], pred_trip[:, [0, 2]]
    keeps = (gt_trip[..., None] == pred_trip.T[None, ...]).all(1)
    out = [[] for _ in range(pred_masks.shape[0])]
    for g in np.where(keeps.any(1))[0]:
        for p in np.where(keeps[g])[0]:
            if phrdet:
                ok = mask_iou(np.logical_or(gt_masks[g, 0], gt_masks[g, 1]),
                              np.logical_or(pred_masks[p, 0], pred_masks[p, 1])) >= thr
            else:
                ok = (mask_iou(gt_masks[g, 0], pred_masks[p, 0]) >= thr and
                      mask_iou(gt_masks[g, 1], pred_masks[p, 1]) >= thr)
            if ok:
                out[p].append(int(g))
    return out


def evaluate(labels, rel_pairs, rel_dists, masks, gt_rels, gt_labels, gt_masks, thr=0.5,
             ks=(20, 50, 100)):
    """numpy inputs as the reference's `Result` / ground truth carry them."""
    pred_rels = np.column_stack((rel_pairs, 1 + rel_dists[:, 1:].argmax(1)))
    gt_trip, gt_tm = triplets(gt_rels, gt_labels, gt_masks)
    p_trip, p_tm = triplets(pred_rels, labels, masks)
    out = {}
    for name, ph in (("sgdet", False), ("phrdet", True)):
        p2g = pred_matches(gt_trip, p_trip, gt_tm, p_tm, thr, phrdet=ph)
        rec = {}
        for k in ks:
            match = reduce(np.union1d, p2g[:k])
            rec[k] = float(len(match)) / float(gt_rels.shape[0])
        out[name + "_recall"] = rec
        out[("" if not ph else "phrdet_") + "pred_to_gt"] = p2g
    return out


# ---- detection_method == "bbox" (sgg_metrics.py:1181-1273) ----
def bbox_overlaps(b1, b2, eps=1e-6):
    """mmdet.core.bbox_overlaps(mode="iou", is_aligned=False) restated (third party, absent
    here): float32 throughout."""
    import torch
    b1, b2 = torch.as_tensor(b1, dtype=torch.float32), torch.as_tensor(b2, dtype=torch.float32)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[None, :, :2])
    rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    ov = wh[..., 0] * wh[..., 1]
    union = torch.max(a1[:, None] + a2[None, :] - ov, ov.new_tensor([eps]))
    return ov / union


def pred_matches_bbox(gt_trip, pred_trip, gt_boxes, pred_boxes, thr, phrdet=False, ignore_rel=False):
    """gt_boxes (G, 8) / pred_boxes (P, 8): subject box | object box per triplet."""
    if ignore_rel:
        gt_trip, pred_trip = gt_trip[:, [0, 2]], pred_trip[:, [0, 2]]
    keeps = (gt_trip[..., None] == pred_trip.T[None, ...]).all(1)
    out = [[] for _ in range(pred_boxes.shape[0])]
    for g in np.where(keeps.any(1))[0]:
        idx = np.where(keeps[g])[0]
        boxes = pred_boxes[idx]
        if phrdet:
            gu = gt_boxes[g].reshape(2, 4)
            gu = np.concatenate((gu.min(0)[:2], gu.max(0)[2:]), 0)
            bu = boxes.reshape(-1, 2, 4)
            bu = np.concatenate((bu.min(1)[:, :2], bu.max(1)[:, 2:]), 1)
            inds = bbox_overlaps(gu[None], bu).numpy()[0] >= thr
        else:
            inds = (bbox_overlaps(gt_boxes[g][None, :4], boxes[:, :4]).numpy()[0] >= thr) & \
                   (bbox_overlaps(gt_boxes[g][None, 4:], boxes[:, 4:]).numpy()[0] >= thr)
        for p in idx[inds]:
            out[p].append(int(g))
    return out


def evaluate_boxes(labels, rel_pairs, rel_dists, boxes, gt_rels, gt_labels, gt_boxes, thr=0.5,
                   ks=(20, 50, 100)):
    pred_rels = np.column_stack((rel_pairs, 1 + rel_dists[:, 1:].argmax(1)))
    trip = lambda rel, cls, bx: (np.column_stack((cls[rel[:, 0]], rel[:, 2], cls[rel[:, 1]])),
                                 np.column_stack((bx[rel[:, 0]], bx[rel[:, 1]])))
    gt_trip, gt_tb = trip(gt_rels, gt_labels, gt_boxes)
    p_trip, p_tb = trip(pred_rels, labels, boxes)
    out = {}
    for name, ph in (("sgdet", False), ("phrdet", True)):
        p2g = pred_matches_bbox(gt_trip, p_trip, gt_tb, p_tb, thr, phrdet=ph)
        rec = {}
        for k in ks:
            match = reduce(np.union1d, p2g[:k])
            rec[k] = float(len(match)) / float(gt_rels.shape[0])
        out[name + "_recall"] = rec
        out[("" if not ph else "phrdet_") + "pred_to_gt"] = p2g
    return out


# ---- dataset-level aggregation: SGRecall / SGMeanRecall / SGPairAccuracy ----------------
# pairnet/evaluation/sgg_metrics.py: recall lists are averaged over images (:100-141,
# `np.mean(v)` in _print_single); SGMeanRecall collects per image, per predicate, the recall
# of that predicate's ground-truth relations (`_collect_single` :741-766; index 0 collects
# "all predicates") and averages per predicate over the images that have it, then over the
# num_rel - 1 predicates (`_calculate_single` :768-792; a predicate no image has counts as
# 0); SGPairAccuracy (:537-667) is a no-op in sgdet mode apart from `prepare_gtpair` (:632-641).
def mean_recall_collect(pred_to_gt, gt_rels, num_rel, ks=(20, 50, 100)):
    """One image -> {k: {predicate n: recall of its gt relations}} (only predicates present)."""
    out = {}
    for k in ks:
        match = reduce(np.union1d, pred_to_gt[:k])
        hit, count = [0] * num_rel, [0] * num_rel
        for idx in range(gt_rels.shape[0]):
            count[int(gt_rels[idx, 2])] += 1
            count[0] += 1
        for idx in range(len(match)):
            hit[int(gt_rels[int(match[idx]), 2])] += 1
            hit[0] += 1
        out[k] = {n: float(hit[n] / count[n]) for n in range(num_rel) if count[n] > 0}
    return out


def mean_recall(collected, num_rel, ks=(20, 50, 100)):
    """`collected`: the per-image dicts of mean_recall_collect -> ({k: mR@k}, {k: per-predicate
    list over predicates 1..num_rel-1})."""
    mr, lists = {}, {}
    for k in ks:
        per = [[] for _ in range(num_rel)]
        for img in collected:
            for n, v in img[k].items():
                per[n].append(v)
        lst = [0.0 if len(per[n + 1]) == 0 else float(np.mean(per[n + 1]))
               for n in range(num_rel - 1)]
        lists[k] = lst
        mr[k] = sum(lst) / float(num_rel - 1)
    return mr, lists


def pred_pair_in_gt(rel_pairs, gt_rels):
    p = rel_pairs[:, 0] * 10000 + rel_pairs[:, 1]
    g = gt_rels[:, 0] * 10000 + gt_rels[:, 1]
    return (p[:, None] == g[None, :]).sum(-1) > 0


def iou_panseg(gt_triplets, pred_classes, gt_triplet_masks, pred_masks):
    """`_compute_iou_panseg` (:1087-1131): for every ground-truth triplet whose subject
    (object) class occurs among the predicted classes, the best mask IoU over the predictions
    of that class."""
    subs, objs = [], []
    for col, which, out in ((0, 0, subs), (2, 1, objs)):
        keep = gt_triplets[:, col][:, None] == pred_classes[None, :]
        for g in np.where(keep.any(1))[0]:
            best = 0
            for pm in pred_masks[keep[g]]:
                best = max(mask_iou(gt_triplet_masks[g, which], pm), best)
            out.append(best)
    return np.array(subs), np.array(objs)
