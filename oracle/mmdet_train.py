"""[3P] training-side pieces the reference's `CrossHead2.loss` calls.

TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import anything under oracle/).  The sources of these classes are NOT under /root/reference:
they belong to mmcv-full 1.7.0 / mmdet 2.25.1 (README.md:80-84), absent from this image.
Every class below restates the published algorithm of the version the reference pins, from
memory -- **parity unpinned** for this file as a whole, with two exceptions pinned in
tests/test_losses.py: `point_sample` and the three matching costs of `MaskHungarianAssigner`
are checked against the same utilities of HuggingFace `transformers`' Mask2Former loss
(`sample_point`, `pair_wise_sigmoid_cross_entropy_loss`, `pair_wise_dice_loss`), an
independent implementation of the same paper.  Reference call sites:

  point_sample                      pairnet_head.py:13, :631-638
  build_assigner / build_sampler    pairnet_head.py:15, :127-134 (cfg pairnet.py:192-208)
  multi_apply                       pairnet_head.py:15, :450, :587
  build_loss                        pairnet_head.py:17, :140-147 (cfg pairnet.py:153-189)
  AssignResult, BaseAssigner,
  build_match_cost                  approaches/matcher.py:3-5 (IdMatcher, :208-275)
  LOSSES, weighted_loss             losses/seg_losses.py:4-5
"""
import functools

import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment


def multi_apply(func, *args, **kwargs):
    """mmdet.core.utils.multi_apply: map, then transpose the tuple of results."""
    pfunc = functools.partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


def point_sample(input, points, align_corners=False, **kwargs):
    """mmcv.ops.point_sample: bilinear samples of `input` (N, C, H, W) at `points`
    (N, P, 2) in [0, 1] x [0, 1] (x, y) -> (N, C, P)."""
    add_dim = False
    if points.dim() == 3:
        add_dim = True
        points = points.unsqueeze(2)
    out = F.grid_sample(input, 2.0 * points - 1.0, align_corners=align_corners, **kwargs)
    if add_dim:
        out = out.squeeze(3)
    return out


class AssignResult:
    """mmdet.core.bbox.assigners.AssignResult (the fields the samplers read)."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


class BaseAssigner:
    pass


class _Registry:
    def __init__(self):
        self.classes = {}

    def register_module(self, *a, **kw):
        def deco(cls):
            self.classes[cls.__name__] = cls
            return cls
        return deco

    def build(self, cfg):
        cfg = dict(cfg)
        return self.classes[cfg.pop("type")](**cfg)


BBOX_ASSIGNERS, MATCH_COST, LOSSES, BBOX_SAMPLERS = _Registry(), _Registry(), _Registry(), _Registry()


# ------------------------------------------------------------------------ match costs
@MATCH_COST.register_module()
class ClassificationCost:
    """mmdet match_cost.ClassificationCost: -softmax(cls_pred)[:, gt] * weight."""

    def __init__(self, weight=1.0):
        self.weight = weight

    def __call__(self, cls_pred, gt_labels):
        cls_score = cls_pred.softmax(-1)
        return -cls_score[:, gt_labels] * self.weight


@MATCH_COST.register_module()
class CrossEntropyLossCost:
    """mmdet match_cost.CrossEntropyLossCost (use_sigmoid=True): mean binary cross entropy
    of every (prediction, target) pair over the sampled points."""

    def __init__(self, weight=1.0, use_sigmoid=True):
        assert use_sigmoid, "use_sigmoid = False is not supported yet."
        self.weight, self.use_sigmoid = weight, use_sigmoid

    def __call__(self, cls_pred, gt_labels):
        cls_pred = cls_pred.flatten(1).float()
        gt_labels = gt_labels.flatten(1).float()
        n = cls_pred.shape[1]
        pos = F.binary_cross_entropy_with_logits(cls_pred, torch.ones_like(cls_pred), reduction="none")
        neg = F.binary_cross_entropy_with_logits(cls_pred, torch.zeros_like(cls_pred), reduction="none")
        cost = torch.einsum("nc,mc->nm", pos, gt_labels) + torch.einsum("nc,mc->nm", neg, 1 - gt_labels)
        return cost / n * self.weight


@MATCH_COST.register_module()
class DiceCost:
    """mmdet match_cost.DiceCost."""

    def __init__(self, weight=1.0, pred_act=False, eps=1e-3, naive_dice=True):
        self.weight, self.pred_act, self.eps, self.naive_dice = weight, pred_act, eps, naive_dice

    def __call__(self, mask_preds, gt_masks):
        if self.pred_act:
            mask_preds = mask_preds.sigmoid()
        mask_preds = mask_preds.flatten(1)
        gt_masks = gt_masks.flatten(1).float()
        numerator = 2 * torch.einsum("nc,mc->nm", mask_preds, gt_masks)
        if self.naive_dice:
            denominator = mask_preds.sum(-1)[:, None] + gt_masks.sum(-1)[None, :]
        else:
            denominator = mask_preds.pow(2).sum(1)[:, None] + gt_masks.pow(2).sum(1)[None, :]
        return (1 - (numerator + self.eps) / (denominator + self.eps)) * self.weight


def build_match_cost(cfg):
    return MATCH_COST.build(cfg)


# -------------------------------------------------------------------- assigner / sampler
@BBOX_ASSIGNERS.register_module()
class MaskHungarianAssigner(BaseAssigner):
    """mmdet MaskHungarianAssigner: one-to-one matching of mask queries and ground truths on
    cls + mask + dice costs (scipy's linear_sum_assignment on the host)."""

    def __init__(self, cls_cost=dict(type="ClassificationCost", weight=1.0),
                 mask_cost=dict(type="FocalLossCost", weight=1.0, binary_input=True),
                 dice_cost=dict(type="DiceCost", weight=1.0)):
        self.cls_cost = build_match_cost(cls_cost)
        self.mask_cost = build_match_cost(mask_cost)
        self.dice_cost = build_match_cost(dice_cost)

    def assign(self, cls_pred, mask_pred, gt_labels, gt_mask, img_meta, gt_bboxes_ignore=None,
               eps=1e-7):
        assert gt_bboxes_ignore is None
        num_gt, num_query = gt_labels.shape[0], cls_pred.shape[0]
        assigned_gt_inds = cls_pred.new_full((num_query,), -1, dtype=torch.long)
        assigned_labels = cls_pred.new_full((num_query,), -1, dtype=torch.long)
        if num_gt == 0 or num_query == 0:
            if num_gt == 0:
                assigned_gt_inds[:] = 0
            return AssignResult(num_gt, assigned_gt_inds, None, labels=assigned_labels)
        cls_cost = self.cls_cost(cls_pred, gt_labels) \
            if self.cls_cost.weight != 0 and cls_pred is not None else 0
        mask_cost = self.mask_cost(mask_pred, gt_mask) if self.mask_cost.weight != 0 else 0
        dice_cost = self.dice_cost(mask_pred, gt_mask) if self.dice_cost.weight != 0 else 0
        cost = (cls_cost + mask_cost + dice_cost).detach().cpu()
        rows, cols = linear_sum_assignment(cost)
        rows = torch.from_numpy(rows).to(cls_pred.device)
        cols = torch.from_numpy(cols).to(cls_pred.device)
        assigned_gt_inds[:] = 0
        assigned_gt_inds[rows] = cols + 1
        assigned_labels[rows] = gt_labels[cols]
        return AssignResult(num_gt, assigned_gt_inds, None, labels=assigned_labels)


class MaskSamplingResult:
    """mmdet MaskSamplingResult (the fields pairnet_head.py:645-692 reads)."""

    def __init__(self, pos_inds, neg_inds, masks, gt_masks, assign_result, gt_flags):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_masks, self.neg_masks = masks[pos_inds], masks[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_masks.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_masks.numel() == 0:
            self.pos_gt_masks = torch.empty_like(gt_masks)
        else:
            self.pos_gt_masks = gt_masks[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None else None


@BBOX_SAMPLERS.register_module()
class MaskPseudoSampler:
    """mmdet MaskPseudoSampler: every assigned query is a positive sample."""

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, masks, gt_masks, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        gt_flags = masks.new_zeros(masks.shape[0], dtype=torch.uint8)
        return MaskSamplingResult(pos_inds, neg_inds, masks, gt_masks, assign_result, gt_flags)


def build_assigner(cfg):
    return BBOX_ASSIGNERS.build(cfg)


def build_sampler(cfg, **default_args):
    cfg = dict(cfg)
    cfg.pop("context", None)
    return BBOX_SAMPLERS.build(cfg)


# ------------------------------------------------------------------------------ losses
def weight_reduce_loss(loss, weight=None, reduction="mean", avg_factor=None):
    """mmdet.models.losses.utils.weight_reduce_loss."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == "mean":
            return loss.mean()
        if reduction == "sum":
            return loss.sum()
        return loss
    if reduction == "mean":
        return loss.sum() / (avg_factor + torch.finfo(torch.float32).eps)
    if reduction == "none":
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


def weighted_loss(loss_func):
    """mmdet.models.losses.utils.weighted_loss (decorator)."""
    @functools.wraps(loss_func)
    def wrapper(pred, target, weight=None, reduction="mean", avg_factor=None, **kwargs):
        return weight_reduce_loss(loss_func(pred, target, **kwargs), weight, reduction, avg_factor)
    return wrapper


def cross_entropy(pred, label, weight=None, reduction="mean", avg_factor=None, class_weight=None,
                  ignore_index=-100, avg_non_ignore=False):
    """mmdet.models.losses.cross_entropy_loss.cross_entropy: the element losses carry the class
    weight, the mean is over ELEMENTS (not torch's weighted mean)."""
    ignore_index = -100 if ignore_index is None else ignore_index
    loss = F.cross_entropy(pred, label, weight=class_weight, reduction="none",
                           ignore_index=ignore_index)
    if avg_factor is None and avg_non_ignore and reduction == "mean":
        avg_factor = label.numel() - (label == ignore_index).sum().item()
    if weight is not None:
        weight = weight.float()
    return weight_reduce_loss(loss, weight=weight, reduction=reduction, avg_factor=avg_factor)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """mmdet CrossEntropyLoss, softmax form (use_sigmoid / use_mask False: what
    `subobj_cls_loss` and `loss_cls` configure, pairnet.py:159-176)."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction="mean", class_weight=None,
                 ignore_index=None, loss_weight=1.0, avg_non_ignore=False):
        super().__init__()
        self.use_sigmoid, self.use_mask = use_sigmoid, use_mask
        self.reduction, self.loss_weight, self.class_weight = reduction, loss_weight, class_weight
        self.ignore_index, self.avg_non_ignore = ignore_index, avg_non_ignore

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None,
                ignore_index=None, **kwargs):
        if self.use_sigmoid or self.use_mask:
            raise NotImplementedError("only the softmax form is on CrossHead2.loss's path")
        reduction = reduction_override if reduction_override else self.reduction
        ignore_index = self.ignore_index if ignore_index is None else ignore_index
        cw = cls_score.new_tensor(self.class_weight) if self.class_weight is not None else None
        return self.loss_weight * cross_entropy(cls_score, label, weight, class_weight=cw,
                                                reduction=reduction, avg_factor=avg_factor,
                                                ignore_index=ignore_index,
                                                avg_non_ignore=self.avg_non_ignore, **kwargs)


@LOSSES.register_module()
class DiceLoss(nn.Module):
    """Built by the reference's constructor (pairnet_head.py:142), never called by `loss`."""

    def __init__(self, **cfg):
        super().__init__()
        self.cfg = cfg


def seesaw_ce_loss(cls_score, labels, label_weights, cum_samples, num_classes, p, q, eps,
                   reduction="mean", avg_factor=None):
    """mmdet.models.losses.seesaw_loss.seesaw_ce_loss (Wang et al., CVPR 2021)."""
    assert cls_score.size(-1) == num_classes
    assert len(cum_samples) == num_classes
    onehot_labels = F.one_hot(labels, num_classes)
    seesaw_weights = cls_score.new_ones(onehot_labels.size())
    if p > 0:     # mitigation factor
        sample_ratio_matrix = cum_samples[None, :].clamp(min=1) / cum_samples[:, None].clamp(min=1)
        index = (sample_ratio_matrix < 1.0).float()
        sample_weights = sample_ratio_matrix.pow(p) * index + (1 - index)
        seesaw_weights = seesaw_weights * sample_weights[labels.long(), :]
    if q > 0:     # compensation factor
        scores = F.softmax(cls_score.detach(), dim=1)
        self_scores = scores[torch.arange(0, len(scores)).to(scores.device).long(), labels.long()]
        score_matrix = scores / self_scores[:, None].clamp(min=eps)
        index = (score_matrix > 1.0).float()
        seesaw_weights = seesaw_weights * (score_matrix.pow(q) * index + (1 - index))
    cls_score = cls_score + (seesaw_weights.log() * (1 - onehot_labels))
    loss = F.cross_entropy(cls_score, labels, weight=None, reduction="none")
    if label_weights is not None:
        label_weights = label_weights.float()
    return weight_reduce_loss(loss, weight=label_weights, reduction=reduction, avg_factor=avg_factor)


@LOSSES.register_module()
class SeesawLoss(nn.Module):
    """mmdet SeesawLoss: C class logits + 2 objectness logits per row; `cum_samples` is a
    persistent buffer that accumulates the label counts of every call."""

    def __init__(self, use_sigmoid=False, p=0.8, q=2.0, num_classes=1203, eps=1e-2,
                 reduction="mean", loss_weight=1.0, return_dict=True):
        super().__init__()
        assert not use_sigmoid
        self.p, self.q, self.num_classes, self.eps = p, q, num_classes, eps
        self.reduction, self.loss_weight, self.return_dict = reduction, loss_weight, return_dict
        self.register_buffer("cum_samples", torch.zeros(self.num_classes + 1, dtype=torch.float))
        self.custom_cls_channels = True
        self.use_sigmoid = False

    def forward(self, cls_score, labels, label_weights=None, avg_factor=None,
                reduction_override=None):
        reduction = reduction_override if reduction_override else self.reduction
        assert cls_score.size(-1) == self.num_classes + 2
        pos_inds = labels < self.num_classes
        obj_labels = (labels == self.num_classes).long()     # 0 for pos, 1 for neg
        for u_l in labels.unique():
            self.cum_samples[u_l] += (labels == u_l.item()).sum()
        if label_weights is not None:
            label_weights = label_weights.float()
        else:
            label_weights = labels.new_ones(labels.size(), dtype=torch.float)
        cls_score_classes, cls_score_objectness = cls_score[..., :-2], cls_score[..., -2:]
        if pos_inds.sum() > 0:
            loss_cls_classes = self.loss_weight * seesaw_ce_loss(
                cls_score_classes[pos_inds], labels[pos_inds], label_weights[pos_inds],
                self.cum_samples[:self.num_classes], self.num_classes, self.p, self.q, self.eps,
                reduction, avg_factor)
        else:
            loss_cls_classes = cls_score_classes[pos_inds].sum()
        loss_cls_objectness = self.loss_weight * cross_entropy(
            cls_score_objectness, obj_labels, label_weights, reduction, avg_factor)
        if self.return_dict:
            return dict(loss_cls_objectness=loss_cls_objectness, loss_cls_classes=loss_cls_classes)
        return loss_cls_classes + loss_cls_objectness


def build_loss(cfg):
    return LOSSES.build(cfg)
