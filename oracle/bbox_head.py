"""CPU restatement of the reference's box-trunk sibling head (TEST INFRASTRUCTURE).

`OracleCrossHeadBBox` restates the inference half of CrossHeadBBox:
  construction         pairnet/models/relation_heads/pairnet_bbox_head.py:22-155
  forward              :193-359
  get_bboxes           :1012-1041
  _get_bboxes_single   :1043-1101
  simple_test_bboxes   :1103-1107
for the configurations that are consistent with the class (two-stage, box-refining
Deformable-DETR trunk, post-norm ReLU relation decoder:
configs/deformable_detr/cross_r101_vg.py:30-117).  The trunk is oracle/deformable_detr.py.
Parameter names are the reference's, so state dicts interchange with the shimmed reference
class; tests/test_oracle.py checks the two bit-for-bit on CPU.

Reference behaviours kept on purpose (they are what the class computes):
  * the 100 kept queries are chosen by `softmax(class logits, dim=1)` -- a softmax over
    the QUERY axis -- then max over classes, then top-100 (:252-254);
  * `sub_pos = idx // 100`, `obj_pos = idx % 100` with the literal 100 (:278-279);
  * `rel_value_pos_embed` reaches the layers as `value_pos`, which mmcv's attention
    ignores (:303-314): dead parameter;
  * `relation_decoder.post_norm` exists and is not applied (:319-320).
"""
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers as L
from .deformable_detr import build_transformer, inverse_sigmoid
from .matrix_learner import build_matrix_learner


def _mlp3(c):
    return nn.Sequential(nn.Linear(c, c), nn.ReLU(inplace=True), nn.Linear(c, c),
                         nn.ReLU(inplace=True), nn.Linear(c, c))


def bbox_cxcywh_to_xyxy(b):
    cx, cy, w, h = b.split((1, 1, 1, 1), dim=-1)
    return torch.cat([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


class OracleCrossHeadBBox(nn.Module):
    def __init__(self, num_classes, num_relations, use_mask=False, num_obj_query=100,
                 num_rel_query=100, transformer=None, embed_dims=256, relation_decoder=None,
                 num_reg_fcs=2, as_two_stage=False, with_box_refine=False,
                 positional_encoding=None, loss_cls=None, **unused):
        super().__init__()
        self.num_classes, self.num_relations = num_classes, num_relations
        self.num_rel_query, self.num_queries = num_rel_query, num_obj_query
        self.use_mask, self.embed_dims = use_mask, embed_dims
        self.as_two_stage, self.with_box_refine = as_two_stage, with_box_refine
        self.relation_decoder = L.build_transformer_layer_sequence(relation_decoder)
        self.rel_query_pos_embed = nn.Embedding(num_rel_query, embed_dims)
        self.rel_key_pos_embed = nn.Embedding(num_rel_query * 2, embed_dims)
        self.rel_value_pos_embed = nn.Embedding(num_rel_query * 2, embed_dims)
        self.rel_query_feat = nn.Embedding(num_rel_query, embed_dims)
        self.update_importance = build_matrix_learner("conv_tiny")
        self.transformer = build_transformer(transformer)
        self.positional_encoding = L.build_positional_encoding(positional_encoding)
        use_sigmoid = bool((loss_cls or {}).get("use_sigmoid", False))
        self.cls_out_channels = num_classes if use_sigmoid else num_classes + 1
        self.sub_query_update = _mlp3(embed_dims)
        self.obj_query_update = _mlp3(embed_dims)
        self.rel_cls_embed = nn.Linear(embed_dims, num_relations)
        fc_cls = nn.Linear(embed_dims, self.cls_out_channels)
        reg = []
        for _ in range(num_reg_fcs):
            reg += [nn.Linear(embed_dims, embed_dims), nn.ReLU()]
        reg = nn.Sequential(*reg, nn.Linear(embed_dims, 4))
        nl = self.transformer.decoder.num_layers
        num_pred = nl + 1 if as_two_stage else nl
        if with_box_refine:
            self.cls_branches = nn.ModuleList(copy.deepcopy(fc_cls) for _ in range(num_pred))
            self.reg_branches = nn.ModuleList(copy.deepcopy(reg) for _ in range(num_pred))
        else:
            self.cls_branches = nn.ModuleList(fc_cls for _ in range(num_pred))
            self.reg_branches = nn.ModuleList(reg for _ in range(num_pred))
        if not as_two_stage:
            self.query_embedding = nn.Embedding(num_obj_query, embed_dims * 2)

    # pairnet_bbox_head.py:193-359
    @torch.no_grad()
    def forward(self, mlvl_feats, img_metas, trace=None):
        bs = mlvl_feats[0].size(0)
        ih, iw = img_metas[0]["batch_input_shape"]
        img_masks = mlvl_feats[0].new_ones((bs, ih, iw))
        for i in range(bs):
            h, w, _ = img_metas[i]["img_shape"]
            img_masks[i, :h, :w] = 0
        masks, poss = [], []
        for feat in mlvl_feats:
            masks.append(F.interpolate(img_masks[None], size=feat.shape[-2:])
                         .to(torch.bool).squeeze(0))
            poss.append(self.positional_encoding(masks[-1]))
        query_embeds = None if self.as_two_stage else self.query_embedding.weight
        hs, init_ref, inter_refs, enc_cls, enc_coord = self.transformer(
            mlvl_feats, masks, query_embeds, poss,
            reg_branches=self.reg_branches if self.with_box_refine else None,
            cls_branches=self.cls_branches if self.as_two_stage else None, trace=trace)
        hs = hs.permute(0, 2, 1, 3)                       # (L, bs, nq, C)
        classes, coords = [], []
        for lvl in range(hs.shape[0]):
            ref = inverse_sigmoid(init_ref if lvl == 0 else inter_refs[lvl - 1])
            classes.append(self.cls_branches[lvl](hs[lvl]))
            tmp = self.reg_branches[lvl](hs[lvl])
            if ref.shape[-1] == 4:
                tmp += ref
            else:
                tmp[..., :2] += ref
            coords.append(tmp.sigmoid())
        classes, coords = torch.stack(classes), torch.stack(coords)
        # softmax over the query axis, as the reference writes it (:252-254)
        query_score = torch.softmax(classes[-1], dim=1).max(-1).values
        index = query_score.topk(100).indices
        outputs_class = torch.gather(classes[-1], 1,
                                     index.unsqueeze(-1).repeat(1, 1, self.num_classes))
        outputs_coord = torch.gather(coords[-1], 1, index.unsqueeze(-1).repeat(1, 1, 4))
        query_feats = hs.clone().transpose(1, 2)          # (L, nq, bs, C)
        query_feats = torch.gather(
            query_feats, 1,
            index.transpose(0, 1).unsqueeze(-1).repeat(hs.shape[0], 1, 1, self.embed_dims))
        sub_embed = self.sub_query_update(query_feats)
        obj_embed = self.obj_query_update(query_feats)
        sub_embed = F.normalize(sub_embed[-1].transpose(0, 1), p=2, dim=-1, eps=1e-12)
        obj_embed = F.normalize(obj_embed[-1].transpose(0, 1), p=2, dim=-1, eps=1e-12)
        importance_raw = torch.matmul(sub_embed, obj_embed.transpose(1, 2))
        importance = self.update_importance(importance_raw)
        _, idx = torch.topk(importance.flatten(-2, -1), k=self.num_rel_query)
        sub_pos = torch.div(idx, 100, rounding_mode="trunc")
        obj_pos = torch.remainder(idx, 100)
        query_feat = query_feats[-1]                      # (100, bs, C)
        take = lambda pos: torch.gather(
            query_feat, 0, pos.unsqueeze(-1).repeat(1, 1, self.embed_dims).transpose(0, 1))
        obj_q, sub_q = take(obj_pos), take(sub_pos)
        rep = lambda e: e.weight.unsqueeze(1).repeat((1, bs, 1))
        rel_q, rel_q_pos = rep(self.rel_query_feat), rep(self.rel_query_pos_embed)
        rel_k_pos, rel_v_pos = rep(self.rel_key_pos_embed), rep(self.rel_value_pos_embed)
        pair_feat = torch.cat([sub_q, obj_q], dim=0)
        for layer in self.relation_decoder.layers:
            rel_q = layer(query=rel_q, key=pair_feat, value=pair_feat, query_pos=rel_q_pos,
                          key_pos=rel_k_pos, value_pos=rel_v_pos,
                          query_key_padding_mask=None, key_padding_mask=None)
        rel_preds = self.rel_cls_embed(rel_q.transpose(0, 1))
        g = lambda t, pos: torch.gather(t, 1, pos.unsqueeze(-1).expand(-1, -1, t.shape[-1]))
        if trace is not None:
            trace.update(hs=hs, classes=classes, coords=coords, query_score=query_score,
                         index=index, query_feat=query_feat, importance_raw=importance_raw,
                         topk_idx=idx, sub_pos=sub_pos, obj_pos=obj_pos, pair_feat=pair_feat)
        all_cls_scores = dict(
            sub=g(outputs_class, sub_pos), obj=g(outputs_class, obj_pos), cls=outputs_class,
            enc_cls_scores=enc_cls,
            enc_bbox_preds=enc_coord.sigmoid() if enc_coord is not None else None,
            rel=rel_preds, importance=importance)
        all_bbox_preds = dict(bbox=outputs_coord, sub_bbox=g(outputs_coord, sub_pos),
                              obj_bbox=g(outputs_coord, obj_pos))
        return all_cls_scores, all_bbox_preds

    # pairnet_bbox_head.py:1012-1041
    def get_bboxes(self, cls_scores, bbox_preds, img_metas, rescale=False):
        out = []
        for i in range(len(img_metas)):
            out.append(self._get_bboxes_single(
                cls_scores["sub"][i], cls_scores["obj"][i], cls_scores["rel"][i],
                bbox_preds["sub_bbox"][i], bbox_preds["obj_bbox"][i],
                img_metas[i]["img_shape"], img_metas[i]["scale_factor"], rescale))
        return out

    # pairnet_bbox_head.py:1043-1101
    def _get_bboxes_single(self, s_cls, o_cls, r_cls, s_box, o_box, img_shape, scale_factor,
                           rescale=False):
        s_scores, s_labels = F.softmax(s_cls, dim=-1).max(-1)
        o_scores, o_labels = F.softmax(o_cls, dim=-1).max(-1)
        s_labels, o_labels = s_labels + 1, o_labels + 1
        r_dists = F.softmax(r_cls, dim=-1).reshape(-1, self.num_relations)
        r_dists = torch.cat([torch.zeros(self.num_rel_query, 1), r_dists], dim=-1)
        labels = torch.cat((s_labels, o_labels), 0)

        def det(box, scores):
            b = bbox_cxcywh_to_xyxy(box)
            b[:, 0::2] = b[:, 0::2] * img_shape[1]
            b[:, 1::2] = b[:, 1::2] * img_shape[0]
            b[:, 0::2].clamp_(min=0, max=img_shape[1])
            b[:, 1::2].clamp_(min=0, max=img_shape[0])
            if rescale:
                b /= b.new_tensor(scale_factor)
            return torch.cat((b, scores.unsqueeze(1)), -1)

        det_bboxes = torch.cat((det(s_box, s_scores), det(o_box, o_scores)), 0)
        rel_pairs = torch.arange(len(det_bboxes), dtype=torch.int).reshape(2, -1).T
        return (det_bboxes, labels, rel_pairs, torch.zeros(100), torch.zeros(100), r_dists)

    # pairnet_bbox_head.py:1103-1107
    def simple_test_bboxes(self, feats, img_metas, rescale=False):
        return self.get_bboxes(*self.forward(feats, img_metas), img_metas, rescale=rescale)
