"""Device-side feed of the PSG evaluator (consumer side).

The reference's `SGRecall.calculate_recall` (pairnet/evaluation/sgg_metrics.py:173-252)
takes a `Result` on the host: it forms the predicted triplets (:207-209, :1292-1308),
matches them with the ground-truth triplets by class equality and mask IoU >= 0.5
(`_compute_pred_matches_panseg`, :1311-1371, `mask_iou` :1374-1380: a Python loop of
`np.count_nonzero` over full-size masks) and counts recall@K over the triplets in query
order (:95-99).  `TripletEvaluator` produces the same `pred_to_gt` lists and recalls from
the device-resident 8-tuple of `get_bboxes`, so the 200 full-size masks never leave the
GPU: masks are bit-packed (`pn_pack_mask_bits`), the IoU counts are exact integer popcounts
(`pn_mask_iou_counts`), the match matrix is one small kernel (`pn_triplet_match`); only
that R x G byte matrix is copied to the host.
"""
import numpy as np
import torch

from . import hip


class TripletEvaluator:
    def __init__(self, iou_thr=0.5, ks=(20, 50, 100)):
        self.iou_thr, self.ks = float(iou_thr), tuple(ks)

    @torch.no_grad()
    @hip.on_device
    def match(self, result, gt_rels, gt_labels, gt_masks, phrdet=False, ignore_rel=False):
        """result: the 8-tuple of `get_bboxes` (device tensors); gt_rels (G, 3) int
        (sub_id, obj_id, predicate); gt_labels (n_obj,) int; gt_masks (n_obj, H, W) bool,
        at the masks' resolution.  Returns the uint8 match matrix [R][G] on the device."""
        labels, masks, r_dists = result[1], result[3], result[7]
        dev = masks.device
        R, C1 = r_dists.shape
        H, W = masks.shape[-2:]
        gt_rels = torch.as_tensor(np.asarray(gt_rels), dtype=torch.int64)
        gt_labels = torch.as_tensor(np.asarray(gt_labels), dtype=torch.int64)
        G, nobj = int(gt_rels.shape[0]), int(gt_labels.shape[0])
        gt_masks = torch.as_tensor(np.asarray(gt_masks)).to(dev).view(nobj, H, W)
        i32 = lambda t: t.to(torch.int32).to(dev).contiguous()
        gtrip = i32(torch.stack([gt_labels[gt_rels[:, 0]], gt_rels[:, 2],
                                 gt_labels[gt_rels[:, 1]]], 1))
        ptrip = torch.empty(R, 3, device=dev, dtype=torch.int32)
        score = torch.empty(R, device=dev, dtype=torch.float32)
        hip.pred_triplets(labels, r_dists, ptrip, score, R, C1)
        nw = (H * W + 63) // 64
        pw = torch.empty(2 * R, nw, device=dev, dtype=torch.int64)
        gw = torch.empty(nobj, nw, device=dev, dtype=torch.int64)
        hip.pack_mask_bits(masks.view(torch.uint8), pw, 2 * R, H * W)
        hip.pack_mask_bits(gt_masks.to(torch.uint8), gw, nobj, H * W)
        ar = torch.arange(R, dtype=torch.int32, device=dev)
        ps, po = ar, ar + R                              # rel_pairs[r] = (r, R + r)
        gs, go = i32(gt_rels[:, 0]), i32(gt_rels[:, 1])
        if phrdet:                                       # union masks (:1343-1350)
            pu = torch.empty(R, nw, device=dev, dtype=torch.int64)
            gu = torch.empty(G, nw, device=dev, dtype=torch.int64)
            hip.mask_or_rows(pw, ps, po, pu, R, nw)
            hip.mask_or_rows(gw, gs, go, gu, G, nw)
            pw, gw, np_, ng = pu, gu, R, G
            ps, gs = ar, torch.arange(G, dtype=torch.int32, device=dev)
            po = go = None
        else:
            np_, ng = 2 * R, nobj
        inter = torch.empty(np_, ng, device=dev, dtype=torch.int32)
        ap = torch.empty(np_, device=dev, dtype=torch.int32)
        ag = torch.empty(ng, device=dev, dtype=torch.int32)
        hip.mask_iou_counts(pw, np_, gw, ng, nw, inter, ap, ag)
        match = torch.empty(R, G, device=dev, dtype=torch.uint8)
        hip.triplet_match(ptrip, gtrip, R, G, inter, ap, ag, ng, ps, po, gs, go, self.iou_thr,
                          phrdet, ignore_rel, match)
        return match

    @torch.no_grad()
    @hip.on_device
    def match_boxes(self, result, gt_rels, gt_labels, gt_boxes, phrdet=False, ignore_rel=False):
        """`detection_method="bbox"`: result = the 6-tuple of `CrossHeadBBox.get_bboxes`
        (det_bboxes [2R,5], labels, rel_pairs, ., ., r_dists); gt_boxes (n_obj, 4) xyxy.
        `_compute_pred_matches_bbox` (sgg_metrics.py:1212-1273) on the device; returns the uint8
        match matrix [R][G]."""
        det, labels, r_dists = result[0], result[1], result[5]
        dev = det.device
        R, C1 = r_dists.shape
        gt_rels = torch.as_tensor(np.asarray(gt_rels), dtype=torch.int64)
        gt_labels = torch.as_tensor(np.asarray(gt_labels), dtype=torch.int64)
        G = int(gt_rels.shape[0])
        gbox = torch.as_tensor(np.asarray(gt_boxes), dtype=torch.float32).to(dev).contiguous()
        i32 = lambda t: t.to(torch.int32).to(dev).contiguous()
        gtrip = i32(torch.stack([gt_labels[gt_rels[:, 0]], gt_rels[:, 2],
                                 gt_labels[gt_rels[:, 1]]], 1))
        ptrip = torch.empty(R, 3, device=dev, dtype=torch.int32)
        score = torch.empty(R, device=dev, dtype=torch.float32)
        hip.pred_triplets(labels, r_dists, ptrip, score, R, C1)
        ar = torch.arange(R, dtype=torch.int32, device=dev)
        match = torch.empty(R, G, device=dev, dtype=torch.uint8)
        hip.triplet_match_boxes(ptrip, gtrip, R, G, det, det.stride(0), gbox, 4, ar, ar + R,
                                i32(gt_rels[:, 0]), i32(gt_rels[:, 1]), self.iou_thr, phrdet,
                                ignore_rel, match)
        return match

    def evaluate_boxes(self, result, gt_rels, gt_labels, gt_boxes):
        """sgdet + phrdet recalls of one image from box results (the graph-constrained part of
        `calculate_recall`, sgg_metrics.py:173-252)."""
        n = len(gt_rels)
        p2g = self.pred_to_gt(self.match_boxes(result, gt_rels, gt_labels, gt_boxes))
        ph = self.pred_to_gt(self.match_boxes(result, gt_rels, gt_labels, gt_boxes, phrdet=True))
        return dict(pred_to_gt=p2g, phrdet_pred_to_gt=ph, sgdet_recall=self.recall(p2g, n),
                    phrdet_recall=self.recall(ph, n))

    def pred_to_gt(self, match):
        """The reference's list of lists (one D2H copy of R x G bytes)."""
        m = match.cpu().numpy().astype(bool)
        return [np.nonzero(row)[0].tolist() for row in m]

    def recall(self, pred_to_gt, num_gt):
        """recall@K over the triplets in query (= top-k) order (:95-99)."""
        out = {}
        for k in self.ks:
            hit = set()
            for lst in pred_to_gt[:k]:
                hit.update(lst)
            out[k] = len(hit) / float(num_gt)
        return out

    def __call__(self, result, gt_rels, gt_labels, gt_masks):
        """sgdet + phrdet recalls of one image, as `calculate_recall` records them."""
        n = len(gt_rels)
        p2g = self.pred_to_gt(self.match(result, gt_rels, gt_labels, gt_masks))
        ph = self.pred_to_gt(self.match(result, gt_rels, gt_labels, gt_masks, phrdet=True))
        return dict(pred_to_gt=p2g, phrdet_pred_to_gt=ph, sgdet_recall=self.recall(p2g, n),
                    phrdet_recall=self.recall(ph, n))
