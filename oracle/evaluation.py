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
