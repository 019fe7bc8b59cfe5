"""CPU restatement of the reference's training-side loss FORWARD for CrossHead2
(first slice: targets and loss values, no backward).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this.  Follows, and is pinned bit for bit (tests/test_losses.py, same inputs, same
torch RNG state) against the reference's own methods executed from /root/reference under
oracle/ref_shim.install_training():

  CrossHead2.loss / loss_single      pairnet/models/relation_heads/pairnet_head.py:419-560
  CrossHead2._get_target_single      pairnet_head.py:614-718
  IdMatcher.assign                   pairnet/models/relation_heads/approaches/matcher.py:208-275
  BCEWithLogitsLoss                  pairnet/models/losses/seg_losses.py:153-166
  PSGTr.forward_train, mask block    pairnet/models/frameworks/psgtr.py:126-141
                                     (`prepare_gt_masks`; that block is two torch calls, F.pad
                                     and F.interpolate(mode="nearest"): pinned against those)

The third-party pieces those methods call (point_sample, MaskHungarianAssigner and its costs,
MaskPseudoSampler, ClassificationCost, SeesawLoss, mmdet's CrossEntropyLoss) are the
restatements of oracle/mmdet_train.py: sources absent from /root/reference, parity unpinned
except where that file says otherwise.

What the reference's `loss` returns is four terms only -- relation classes (Seesaw), subject
and object classes (cross entropy), and the importance-matrix match (BCE with pos_weight); the
Mask2Former object losses (`loss_cls`, `loss_mask`, `loss_dice`) are built in the constructor
and never called (pairnet_head.py:467-477).
"""
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

from . import mmdet_train as T


def prepare_gt_masks(masks, H, W):
    """psgtr.py:126-141: one image's ground-truth masks [G, h, w] (uint8 0/1) zero-padded on
    the right / bottom to the batch tensor's (H, W), then nearest-neighbour resized to
    (H // 2, W // 2).  ATen's legacy "nearest" reads source index
    min(floor(dst * float32(in / out)), in - 1), the product in float32."""
    import numpy as np
    m = np.asarray(masks)
    G, h, w = m.shape
    Ho, Wo = H // 2, W // 2
    padded = np.zeros((G, H, W), dtype=m.dtype)
    padded[:, :h, :w] = m
    sy = np.float32(H) / np.float32(Ho)
    sx = np.float32(W) / np.float32(Wo)
    iy = np.minimum(np.floor(np.arange(Ho, dtype=np.float32) * sy).astype(np.int64), H - 1)
    ix = np.minimum(np.floor(np.arange(Wo, dtype=np.float32) * sx).astype(np.int64), W - 1)
    return padded[:, iy[:, None], ix[None, :]]


class OracleCrossHead2Loss:
    def __init__(self, num_obj_query=100, num_rel_query=100, num_relations=56, num_points=12544,
                 mask_assigner=None, id_costs=(1.0, 1.0, 0.0), rel_loss=None,
                 subobj_loss_weight=4.0, subobj_class_weight=None, match_loss_weight=5.0):
        self.Q, self.R, self.C = num_obj_query, num_rel_query, num_relations
        self.num_points = num_points
        self.mask_assigner = mask_assigner or T.MaskHungarianAssigner(
            cls_cost=dict(type="ClassificationCost", weight=2.0),
            mask_cost=dict(type="CrossEntropyLossCost", weight=5.0, use_sigmoid=True),
            dice_cost=dict(type="DiceCost", weight=5.0, pred_act=True, eps=1.0))
        self.sampler = T.MaskPseudoSampler()
        self.id_costs = id_costs                      # (subject, object, relation) weights
        self.rel_loss = rel_loss or T.SeesawLoss(num_classes=num_relations, return_dict=True,
                                                 loss_weight=2.0)
        self.subobj_w, self.subobj_cw = subobj_loss_weight, subobj_class_weight
        self.match_w = match_loss_weight

    @classmethod
    def from_config(cls, model_cfg):
        """`model` dict of configs/mask2former/pairnet.py (bbox_head + train_cfg)."""
        h, t = model_cfg["bbox_head"], model_cfg["train_cfg"]
        ida = t["id_assigner"]
        rl = dict(h["rel_cls_loss"])
        assert rl.pop("type") == "SeesawLoss" and h["subobj_cls_loss"]["type"] == "CrossEntropyLoss"
        ma = dict(t["mask_assigner"])
        ma.pop("type")
        return cls(h["num_obj_query"], h["num_rel_query"], h["num_relations"],
                   t.get("num_points", 12544), T.MaskHungarianAssigner(**ma),
                   (ida["sub_id_cost"]["weight"], ida["obj_id_cost"]["weight"],
                    ida["r_cls_cost"]["weight"]), T.SeesawLoss(**rl),
                   h["subobj_cls_loss"]["loss_weight"], h["subobj_cls_loss"].get("class_weight"),
                   h["importance_match_loss"]["loss_weight"])

    # ---- IdMatcher (matcher.py:208-275): Hungarian on -softmax class costs ----
    def id_match(self, sub_score, obj_score, rel_score, gt_sub, gt_obj, gt_rel):
        num_gts, n = gt_rel.shape[0], rel_score.shape[0]
        gt_inds = rel_score.new_full((n,), -1, dtype=torch.long)
        if num_gts == 0 or n == 0:
            if num_gts == 0:
                gt_inds[:] = 0
            return gt_inds
        ws, wo, wr = self.id_costs
        cost = (-sub_score.softmax(-1)[:, gt_sub] * ws) + (-obj_score.softmax(-1)[:, gt_obj] * wo) \
            + (-rel_score.softmax(-1)[:, gt_rel] * wr)
        rows, cols = linear_sum_assignment(cost.detach().cpu())
        gt_inds[:] = 0
        gt_inds[torch.from_numpy(rows)] = torch.from_numpy(cols) + 1
        return gt_inds

    # ---- _get_target_single (pairnet_head.py:614-718) ----
    def targets_single(self, sub_score, obj_score, cls_score, mask_pred, rel_score, gt_rels,
                       gt_labels, gt_masks, point_coords=None, trace=None):
        num_gts = gt_labels.shape[0]
        if gt_rels.shape[0] == 0:
            # the reference fails here too (its IdMatcher returns a pair of results for an
            # empty ground truth, which the sampler cannot read: an AttributeError)
            raise ValueError("an image without ground-truth relations cannot be a loss target "
                             "(the reference's CrossHead2.loss fails on it as well)")
        if point_coords is None:
            point_coords = torch.rand((1, self.num_points, 2), device=cls_score.device)
        pred_pts = T.point_sample(mask_pred.unsqueeze(1), point_coords.repeat(self.Q, 1, 1)).squeeze(1)
        gt_pts = T.point_sample(gt_masks.unsqueeze(1).float(),
                                point_coords.repeat(num_gts, 1, 1)).squeeze(1)
        assign = self.mask_assigner.assign(cls_score, pred_pts, gt_labels, gt_pts, None)
        pos_q = torch.nonzero(assign.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        pos_gt = assign.gt_inds[pos_q] - 1
        # ground-truth object -> the object query it was matched to (unmatched: 1, the
        # reference's torch.ones_like initialisation, pairnet_head.py:648)
        query_of_gt = torch.ones_like(gt_labels)
        query_of_gt[pos_gt] = pos_q
        rels = gt_rels.T.long()
        gt_rel = rels[2] - 1
        gt_sub_cls, gt_obj_cls = gt_labels[rels[0]], gt_labels[rels[1]]
        importance = torch.zeros((self.Q, self.Q), device=gt_labels.device)
        importance[query_of_gt[rels[0]], query_of_gt[rels[1]]] += 1     # (no accumulation of
        # duplicates: index_put without accumulate, as in the reference :660)
        tri = self.id_match(sub_score, obj_score, rel_score, gt_sub_cls, gt_obj_cls, gt_rel)
        pos = torch.nonzero(tri > 0, as_tuple=False).squeeze(-1).unique()
        which = tri[pos] - 1
        full = lambda: torch.full((self.R,), -1, dtype=torch.long, device=gt_labels.device)
        sub_ids, obj_ids, r_labels = full(), full(), full()
        sub_ids[pos], obj_ids[pos], r_labels[pos] = gt_sub_cls[which], gt_obj_cls[which], gt_rel[which]
        weights = gt_labels.new_zeros(self.R)
        weights[pos] = 1.0
        if trace is not None:
            trace.update(point_coords=point_coords, pred_pts=pred_pts, gt_pts=gt_pts,
                         mask_gt_inds=assign.gt_inds, triplet_gt_inds=tri)
        return r_labels, weights, sub_ids, obj_ids, importance

    # ---- loss / loss_single (pairnet_head.py:419-560) ----
    def loss(self, cls_scores, mask_preds, gt_rels_list, gt_labels_list, gt_masks_list,
             point_coords=None, trace=None):
        B = cls_scores["cls"].size(0)
        per = []
        for i in range(B):
            tr = {} if trace is not None else None
            per.append(self.targets_single(
                cls_scores["sub"][i], cls_scores["obj"][i], cls_scores["cls"][i],
                mask_preds["mask"][i], cls_scores["rel"][i], gt_rels_list[i], gt_labels_list[i],
                gt_masks_list[i], None if point_coords is None else point_coords[i], tr))
            if trace is not None:
                trace.setdefault("images", []).append(tr)
        r_labels, weights, sub_ids, obj_ids, importance = (list(x) for x in zip(*per))
        keep = torch.cat(weights, 0) > 0
        ce = lambda pred, tgt: self.subobj_w * T.cross_entropy(
            pred, tgt, class_weight=None if self.subobj_cw is None else pred.new_tensor(self.subobj_cw))
        loss_obj = ce(cls_scores["obj"].flatten(0, 1)[keep], torch.cat(obj_ids, 0)[keep])
        loss_sub = ce(cls_scores["sub"].flatten(0, 1)[keep], torch.cat(sub_ids, 0)[keep])
        rel = cls_scores["rel"].reshape(-1, self.C)
        dummy = torch.zeros((int(keep.sum()), 2)).to(rel.device)
        loss_rel = self.rel_loss(torch.cat([rel[keep], dummy], dim=1),
                                 torch.cat(r_labels, 0)[keep])["loss_cls_classes"]
        gt_imp = torch.stack(importance, 0)
        pos_weight = torch.numel(gt_imp) / (gt_imp > 0).sum()
        loss_match = F.binary_cross_entropy_with_logits(
            cls_scores["importance"], gt_imp, pos_weight=pos_weight, reduction="mean") * self.match_w
        if trace is not None:
            trace.update(r_labels=r_labels, weights=weights, sub_ids=sub_ids, obj_ids=obj_ids,
                         importance=gt_imp, pos_weight=pos_weight)
        return dict(loss_r_cls=loss_rel, loss_sub_cls=loss_sub, loss_obj_cls=loss_obj,
                    loss_match=loss_match)
