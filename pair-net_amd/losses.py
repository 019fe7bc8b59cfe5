"""Loss FORWARD of `CrossHead2` on device outputs (first slice).

`CrossHead2Loss.loss(...)` takes what the reference's `CrossHead2.loss` takes
(pairnet/models/relation_heads/pairnet_head.py:419-430) -- the two output dicts of `forward`
and the per-image ground truth -- and returns the same four terms (`loss_r_cls`,
`loss_sub_cls`, `loss_obj_cls`, `loss_match`, :470-477) as 0-dim device tensors: the VALUES
(validation losses) and, with `grads={}`, the gradients of their sum with respect to the four
logit tensors they are computed from (round 6: where the backward pass of grad.py starts;
train.py runs the whole iteration).

Where the arithmetic runs (csrc/loss.hip, one small kernel each):
  pn_point_sample_f32       [3P] mmcv point_sample of the Q mask logit maps and the ground-truth
                            masks at the image's random points (pairnet_head.py:630-638)
  pn_mask_match_cost_f32    [3P] mmdet MaskHungarianAssigner costs (cfg pairnet.py:200-206)
  pn_id_match_cost_f32      IdMatcher costs (approaches/matcher.py:250-258)
  pn_ce_mean_f32            [3P] mmdet CrossEntropyLoss for subjects / objects (:518-527)
  pn_seesaw_mean_f32        [3P] mmdet SeesawLoss, class part, for relations (:529-536)
  pn_bce_posw_mean_f32      BCEWithLogitsLoss with pos_weight (seg_losses.py:153-166; :541-552)
The two Hungarian assignments are solved on the host with scipy, exactly where the reference
solves them (`linear_sum_assignment(cost.cpu())`, matcher.py:262-264): the Q x G and R x G cost
matrices are the only D2H copies; the index bookkeeping of `_get_target_single` (:645-718:
a few dozen integers per image) is done on the host from the assignment and uploaded as the
target vectors.

Reference quirks kept (tests/test_losses_gpu.py): unmatched ground-truth objects point at
query 1 (`torch.ones_like`, :648); duplicate (subject query, object query) pairs set the
importance target to 1, not to their count (:660); an image without ground-truth relations
is an error (the reference fails with an AttributeError there, see oracle/losses.py).
"""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from . import hip
from .config import ConfigDict


class CrossHead2Loss:
    def __init__(self, num_classes, num_relations, num_obj_query=100, num_rel_query=100,
                 train_cfg=None, rel_cls_loss=None, subobj_cls_loss=None,
                 importance_match_loss=None):
        t = ConfigDict(train_cfg or dict(
            id_assigner=dict(type="IdMatcher", sub_id_cost=dict(type="ClassificationCost", weight=1.0),
                             obj_id_cost=dict(type="ClassificationCost", weight=1.0),
                             r_cls_cost=dict(type="ClassificationCost", weight=0.0)),
            num_points=12544,
            mask_assigner=dict(type="MaskHungarianAssigner",
                               cls_cost=dict(type="ClassificationCost", weight=2.0),
                               mask_cost=dict(type="CrossEntropyLossCost", weight=5.0, use_sigmoid=True),
                               dice_cost=dict(type="DiceCost", weight=5.0, pred_act=True, eps=1.0)),
            sampler=dict(type="MaskPseudoSampler")))
        rel = dict(rel_cls_loss or dict(type="SeesawLoss", num_classes=num_relations,
                                        return_dict=True, loss_weight=2.0))
        so = dict(subobj_cls_loss or dict(type="CrossEntropyLoss", use_sigmoid=False,
                                          loss_weight=4.0, reduction="mean"))
        im = dict(importance_match_loss or dict(type="BCEWithLogitsLoss", reduction="mean",
                                                loss_weight=5.0))
        ma, ida = t["mask_assigner"], t["id_assigner"]
        want = [(ma["type"], "MaskHungarianAssigner"), (ida["type"], "IdMatcher"),
                (ma["cls_cost"]["type"], "ClassificationCost"),
                (ma["mask_cost"]["type"], "CrossEntropyLossCost"), (ma["dice_cost"]["type"], "DiceCost"),
                (t.get("sampler", dict(type="MaskPseudoSampler"))["type"], "MaskPseudoSampler"),
                (rel["type"], "SeesawLoss"), (so["type"], "CrossEntropyLoss"),
                (im["type"], "BCEWithLogitsLoss")]
        for got, exp in want:
            if got != exp:
                raise NotImplementedError("%s (built: %s, configs/mask2former/pairnet.py:153-208)"
                                          % (got, exp))
        if not ma["mask_cost"].get("use_sigmoid", True) or not ma["dice_cost"].get("pred_act", False) \
                or not ma["dice_cost"].get("naive_dice", True) or so.get("use_sigmoid", False) \
                or so.get("reduction", "mean") != "mean" or im.get("reduction", "mean") != "mean" \
                or rel.get("reduction", "mean") != "mean":
            raise NotImplementedError("loss options outside configs/mask2former/pairnet.py")
        self.Q, self.R = num_obj_query, num_rel_query
        self.num_classes, self.num_relations = num_classes, num_relations
        self.num_points = int(t.get("num_points", 12544))
        self.w_cls, self.w_mask = float(ma["cls_cost"]["weight"]), float(ma["mask_cost"]["weight"])
        self.w_dice, self.dice_eps = float(ma["dice_cost"]["weight"]), float(ma["dice_cost"].get("eps", 1e-3))
        self.id_w = tuple(float(ida[k]["weight"]) for k in ("sub_id_cost", "obj_id_cost", "r_cls_cost"))
        self.seesaw = dict(p=float(rel.get("p", 0.8)), q=float(rel.get("q", 2.0)),
                           eps=float(rel.get("eps", 1e-2)), loss_weight=float(rel.get("loss_weight", 1.0)))
        if int(rel.get("num_classes", num_relations)) != num_relations or num_relations > 64:
            raise NotImplementedError("SeesawLoss over num_relations <= 64 classes")
        # SeesawLoss.cum_samples: the persistent label counts (num_classes + 1 slots, the last
        # one for the objectness dummy the reference appends and never labels)
        self.cum_samples = np.zeros(num_relations + 1, dtype=np.float32)
        self.subobj_w = float(so.get("loss_weight", 1.0))
        self.subobj_cw = so.get("class_weight")
        self.match_w = float(im.get("loss_weight", 1.0))
        self._cw = None

    def state_dict(self):
        return {"rel_cls_loss.cum_samples": torch.from_numpy(self.cum_samples.copy())}

    def load_state_dict(self, sd):
        self.cum_samples = sd["rel_cls_loss.cum_samples"].float().cpu().numpy().copy()

    # ---- _get_target_single (pairnet_head.py:614-718) ----
    def _targets_single(self, sub, obj, cls, mask_pred, rel, gt_rels, gt_labels, gt_masks,
                        point_coords, trace):
        dev = cls.device
        gt_rels = np.asarray(torch.as_tensor(gt_rels).cpu()).reshape(-1, 3).astype(np.int64)
        gl = np.asarray(torch.as_tensor(gt_labels).cpu()).astype(np.int64)
        if gt_rels.shape[0] == 0:
            raise ValueError("an image without ground-truth relations cannot be a loss target "
                             "(the reference's CrossHead2.loss fails on it as well)")
        G = gl.shape[0]
        if point_coords is None:
            point_coords = torch.rand((1, self.num_points, 2), device=dev)
        pts = point_coords.reshape(-1, 2).to(dev, torch.float32).contiguous()
        Np = pts.shape[0]
        f32 = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        pred_pts, gt_pts = f32(self.Q, Np), f32(G, Np)
        hip.point_sample(mask_pred.contiguous(), pts, pred_pts)
        gm = torch.as_tensor(gt_masks).to(dev)
        if gm.dtype not in (torch.bool, torch.uint8):
            gm = gm.to(torch.float32)
        hip.point_sample(gm.contiguous(), pts, gt_pts)
        cost = f32(self.Q, G)
        gl_dev = torch.from_numpy(gl).to(dev)
        hip.mask_match_cost(cls.contiguous(), gl_dev, pred_pts, gt_pts, cost, self.w_cls,
                            self.w_mask, self.w_dice, self.dice_eps)
        rows, cols = linear_sum_assignment(cost.cpu().numpy())          # (host, as the reference)
        # ground-truth object -> its matched object query; unmatched ones keep the reference's 1
        query_of_gt = np.ones(G, dtype=np.int64)
        order = np.argsort(rows)        # MaskPseudoSampler: positives in ascending query order
        query_of_gt[cols[order]] = rows[order]
        gt_rel = gt_rels[:, 2] - 1
        gt_sub_cls, gt_obj_cls = gl[gt_rels[:, 0]], gl[gt_rels[:, 1]]
        importance = np.zeros((self.Q, self.Q), dtype=np.float32)
        importance[query_of_gt[gt_rels[:, 0]], query_of_gt[gt_rels[:, 1]]] = 1.0
        T = gt_rels.shape[0]
        cost2 = f32(self.R, T)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        hip.id_match_cost(sub.contiguous(), obj.contiguous(), rel.contiguous(), up(gt_sub_cls),
                          up(gt_obj_cls), up(gt_rel), cost2, *self.id_w)
        rows2, cols2 = linear_sum_assignment(cost2.cpu().numpy())
        r_labels = np.full(self.R, -1, dtype=np.int64)
        sub_ids, obj_ids = r_labels.copy(), r_labels.copy()
        r_labels[rows2], sub_ids[rows2], obj_ids[rows2] = gt_rel[cols2], gt_sub_cls[cols2], gt_obj_cls[cols2]
        if trace is not None:
            trace.append(dict(point_coords=point_coords, mask_rows=rows, mask_cols=cols,
                              triplet_rows=rows2, triplet_cols=cols2, mask_cost=cost.cpu(),
                              id_cost=cost2.cpu(), pred_pts=pred_pts, gt_pts=gt_pts))
        return r_labels, sub_ids, obj_ids, importance

    # ---- loss / loss_single (pairnet_head.py:419-560) ----
    @torch.no_grad()
    @hip.on_device
    def loss(self, all_cls_scores, all_mask_preds, gt_rels_list, gt_bboxes_list, gt_labels_list,
             gt_masks_list, img_metas, gt_bboxes_ignore=None, point_coords=None, trace=None,
             grads=None):
        """`point_coords`: optional list of (1, num_points, 2) tensors, one per image (default:
        `torch.rand` on the device, one draw per image in image order, as the reference)."""
        assert gt_bboxes_ignore is None, "Only supports for gt_bboxes_ignore setting to None."
        cls, sub, obj = all_cls_scores["cls"], all_cls_scores["sub"], all_cls_scores["obj"]
        rel, imp, masks = all_cls_scores["rel"], all_cls_scores["importance"], all_mask_preds["mask"]
        B, dev = cls.shape[0], cls.device
        r_lab, s_ids, o_ids, gt_imp = [], [], [], []
        for i in range(B):
            out = self._targets_single(sub[i], obj[i], cls[i], masks[i], rel[i], gt_rels_list[i],
                                       gt_labels_list[i], gt_masks_list[i],
                                       None if point_coords is None else point_coords[i], trace)
            for lst, v in zip((r_lab, s_ids, o_ids, gt_imp), out):
                lst.append(v)
        r_lab, s_ids, o_ids = (np.concatenate(x) for x in (r_lab, s_ids, o_ids))
        # SeesawLoss accumulates this batch's labels before it weighs (seesaw_loss.py forward)
        kept = r_lab[r_lab >= 0]
        np.add.at(self.cum_samples, kept, 1.0)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        out = torch.empty(6, device=dev, dtype=torch.float32)
        if self.subobj_cw is not None and (self._cw is None or self._cw.device != dev):
            self._cw = torch.tensor(self.subobj_cw, dtype=torch.float32, device=dev)
        nc = sub.shape[-1]
        if self._cw is not None and self._cw.numel() != nc:
            raise ValueError("subobj_cls_loss.class_weight has %d entries for %d class logits"
                             % (self._cw.numel(), nc))
        hip.ce_mean(obj.reshape(-1, nc), up(o_ids), self._cw, out[0:1], self.subobj_w)
        hip.ce_mean(sub.reshape(-1, nc), up(s_ids), self._cw, out[1:2], self.subobj_w)
        hip.seesaw_mean(rel.reshape(-1, self.num_relations), up(r_lab),
                        up(self.cum_samples[:self.num_relations]), out[2:3],
                        self.seesaw["p"], self.seesaw["q"], self.seesaw["eps"],
                        self.seesaw["loss_weight"])
        t_imp = up(np.stack(gt_imp, 0))
        hip.bce_posw_mean(imp.contiguous(), t_imp, out[4:6], self.match_w)
        if grads is not None:
            # SURVEY 8 f-4, first backward slice: d (sum of the four terms) / d their logits, by the
            # analytic derivative kernels beside each reduction (csrc/loss.hip); every term depends
            # on its own logits only.  Filled in place: {"obj", "sub", "rel", "importance"}.
            t_o, t_s, t_r = up(o_ids), up(s_ids), up(r_lab)
            cum = up(self.cum_samples[:self.num_relations])
            g = {k: torch.empty_like(v, memory_format=torch.contiguous_format)
                 for k, v in (("obj", obj), ("sub", sub), ("rel", rel), ("importance", imp))}
            hip.ce_mean_grad(obj.reshape(-1, nc), t_o, self._cw, g["obj"].view(-1, nc), self.subobj_w)
            hip.ce_mean_grad(sub.reshape(-1, nc), t_s, self._cw, g["sub"].view(-1, nc), self.subobj_w)
            hip.seesaw_mean_grad(rel.reshape(-1, self.num_relations), t_r, cum,
                                 g["rel"].view(-1, self.num_relations), self.seesaw["p"],
                                 self.seesaw["q"], self.seesaw["eps"], self.seesaw["loss_weight"])
            hip.bce_posw_mean_grad(imp.contiguous(), t_imp, g["importance"], self.match_w)
            grads.update(g)
        return dict(loss_r_cls=out[2], loss_sub_cls=out[1], loss_obj_cls=out[0], loss_match=out[4])
