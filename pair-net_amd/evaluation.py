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
that R x G byte matrix is copied to the host.  `SceneGraphMetrics` is the dataset-level
aggregation on top: R@K averaged over images (:100-141), the reference's headline metric
mean recall (`SGMeanRecall`, :669-916: per-predicate recall averaged over the images that
have the predicate, then over the predicates), phrase-detection variants, and the
subject / object IoU statistic (`_compute_iou_panseg`, :1087-1131, from the same exact
popcounts).  Images without ground-truth relations are skipped, as the reference's
`sgg_evaluate` loop does.
"""
import numpy as np
import torch

from . import hip


def _tensor(x):
    """Ground truth as handed in: numpy, lists, or tensors (host or ALREADY on the device, as
    `pairnet_amd.dataset.eval_ground_truth` leaves the masks)."""
    return x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))


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
        gt_rels = _tensor(gt_rels).cpu().to(torch.int64)
        gt_labels = _tensor(gt_labels).cpu().to(torch.int64)
        G, nobj = int(gt_rels.shape[0]), int(gt_labels.shape[0])
        gt_masks = _tensor(gt_masks).to(dev).view(nobj, H, W)
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
        gt_rels = _tensor(gt_rels).cpu().to(torch.int64)
        gt_labels = _tensor(gt_labels).cpu().to(torch.int64)
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
        if n == 0:
            R = int(result[5].shape[0])
            return dict(pred_to_gt=[[] for _ in range(R)], phrdet_pred_to_gt=[[] for _ in range(R)],
                        sgdet_recall=None, phrdet_recall=None)
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

    @torch.no_grad()
    @hip.on_device
    def iou_stats(self, result, gt_rels, gt_labels, gt_masks):
        """`_compute_iou_panseg` (sgg_metrics.py:1087-1131) from device-resident masks: for
        every ground-truth triplet whose subject (object) class occurs among the 2R predicted
        labels, the best mask IoU over the predictions of that class.  Returns two float64
        arrays (subjects, objects), in ground-truth triplet order."""
        labels, masks = result[1], result[3]
        dev = masks.device
        H, W = masks.shape[-2:]
        gt_rels = np.asarray(gt_rels)
        gt_labels_np = np.asarray(gt_labels)
        nobj, P = int(gt_labels_np.shape[0]), int(masks.shape[0])
        gm = _tensor(gt_masks).to(dev).view(nobj, H, W)
        nw = (H * W + 63) // 64
        pw = torch.empty(P, nw, device=dev, dtype=torch.int64)
        gw = torch.empty(nobj, nw, device=dev, dtype=torch.int64)
        hip.pack_mask_bits(masks.view(torch.uint8), pw, P, H * W)
        hip.pack_mask_bits(gm.to(torch.uint8), gw, nobj, H * W)
        inter = torch.empty(P, nobj, device=dev, dtype=torch.int32)
        ap = torch.empty(P, device=dev, dtype=torch.int32)
        ag = torch.empty(nobj, device=dev, dtype=torch.int32)
        hip.mask_iou_counts(pw, P, gw, nobj, nw, inter, ap, ag)
        inter, ap, ag = (t.cpu().numpy().astype(np.int64) for t in (inter, ap, ag))
        with np.errstate(invalid="ignore", divide="ignore"):
            iou = inter.astype(np.float64) / (ap[:, None] + ag[None, :] - inter).astype(np.float64)
        pl = labels.cpu().numpy()
        out = []
        for col in (0, 1):
            vals = []
            for g in range(gt_rels.shape[0]):
                o = int(gt_rels[g, col])
                same = pl == gt_labels_np[o]
                if same.any():
                    best = 0
                    for v in iou[same, o]:      # (python max(v, best) with the reference's argument
                        # order: a NaN IoU -- empty prediction AND empty ground truth -- DOES
                        # replace `best`, max(nan, x) returns nan; kept for parity)
                        best = max(v, best)
                    vals.append(best)
            out.append(np.array(vals))
        return out[0], out[1]

    def __call__(self, result, gt_rels, gt_labels, gt_masks):
        """sgdet + phrdet recalls of one image, as `calculate_recall` records them.  An image
        without ground-truth relations has nothing to match (the reference's loop skips it):
        empty lists, no recalls."""
        n = len(gt_rels)
        if n == 0:
            R = int(result[7].shape[0])
            return dict(pred_to_gt=[[] for _ in range(R)], phrdet_pred_to_gt=[[] for _ in range(R)],
                        sgdet_recall=None, phrdet_recall=None)
        p2g = self.pred_to_gt(self.match(result, gt_rels, gt_labels, gt_masks))
        ph = self.pred_to_gt(self.match(result, gt_rels, gt_labels, gt_masks, phrdet=True))
        return dict(pred_to_gt=p2g, phrdet_pred_to_gt=ph, sgdet_recall=self.recall(p2g, n),
                    phrdet_recall=self.recall(ph, n))


class SceneGraphMetrics:
    """Dataset-level aggregation of `TripletEvaluator` outputs: what the reference's
    `sgg_evaluate` prints for mode "sgdet" -- R@K and mR@K (graph constraint), their
    phrase-detection variants, and the subject / object IoU lists.

        metrics = SceneGraphMetrics(num_predicates=56)
        for each image:  metrics.add(evaluator(result, gt_rels, gt_labels, gt_masks), gt_rels)
        metrics.summary()   ->  {"sgdet_recall": {20: .., 50: .., 100: ..},
                                 "sgdet_mean_recall": {...}, "sgdet_mean_recall_list": {...},
                                 "phrdet_recall": ..., "phrdet_mean_recall": ..., "images": n}
    """

    def __init__(self, num_predicates, ks=(20, 50, 100)):
        self.num_rel = int(num_predicates) + 1        # + __background__ (:681-683)
        self.ks = tuple(ks)
        self.recalls = {m: {k: [] for k in self.ks} for m in ("sgdet", "phrdet")}
        self.collect = {m: {k: [[] for _ in range(self.num_rel)] for k in self.ks}
                        for m in ("sgdet", "phrdet")}
        self.sub_iou, self.obj_iou = [], []
        self.images = self.skipped = 0

    def _collect(self, mode, pred_to_gt, gt_rels):
        """`SGMeanRecall._collect_single` (:741-766)."""
        for k in self.ks:
            match = set()
            for lst in pred_to_gt[:k]:
                match.update(lst)
            hit, count = [0] * self.num_rel, [0] * self.num_rel
            for g in range(gt_rels.shape[0]):
                count[int(gt_rels[g, 2])] += 1
                count[0] += 1
            for g in match:
                hit[int(gt_rels[int(g), 2])] += 1
                hit[0] += 1
            for n in range(self.num_rel):
                if count[n] > 0:
                    self.collect[mode][k][n].append(float(hit[n] / count[n]))

    def add(self, image_eval, gt_rels, iou=None):
        """`image_eval`: what `TripletEvaluator.__call__` / `evaluate_boxes` returned for the
        image; `iou`: optionally `TripletEvaluator.iou_stats(...)` of the same image."""
        gt_rels = np.asarray(gt_rels).reshape(-1, 3)
        if gt_rels.shape[0] == 0:
            self.skipped += 1
            return
        self.images += 1
        for mode, key in (("sgdet", "pred_to_gt"), ("phrdet", "phrdet_pred_to_gt")):
            for k in self.ks:
                self.recalls[mode][k].append(image_eval[mode + "_recall"][k])
            self._collect(mode, image_eval[key], gt_rels)
        if iou is not None:
            self.sub_iou.extend(iou[0])
            self.obj_iou.extend(iou[1])

    def summary(self):
        """`SGRecall._print_single` / `SGMeanRecall._calculate_single` (:768-792)."""
        out = dict(images=self.images, skipped=self.skipped)
        for mode in ("sgdet", "phrdet"):
            out[mode + "_recall"] = {k: float(np.mean(v)) if v else 0.0
                                     for k, v in self.recalls[mode].items()}
            mr, lists = {}, {}
            for k in self.ks:
                per = self.collect[mode][k]
                lst = [0.0 if len(per[n + 1]) == 0 else float(np.mean(per[n + 1]))
                       for n in range(self.num_rel - 1)]
                lists[k] = lst
                mr[k] = sum(lst) / float(self.num_rel - 1)
            out[mode + "_mean_recall"], out[mode + "_mean_recall_list"] = mr, lists
        if self.sub_iou or self.obj_iou:
            out["subject-IoU"] = float(np.mean(self.sub_iou)) if self.sub_iou else 0.0
            out["object-IoU"] = float(np.mean(self.obj_iou)) if self.obj_iou else 0.0
        return out

    @staticmethod
    def pred_pair_in_gt(rel_pairs, gt_rels):
        """`SGPairAccuracy.prepare_gtpair` (:632-641).  (The accuracy itself is only
        accumulated for modes other than "sgdet", :571-585: nothing to add for PSG.)"""
        rel_pairs, gt_rels = np.asarray(rel_pairs), np.asarray(gt_rels)
        p = rel_pairs[:, 0] * 10000 + rel_pairs[:, 1]
        g = gt_rels[:, 0] * 10000 + gt_rels[:, 1]
        return (p[:, None] == g[None, :]).sum(-1) > 0
