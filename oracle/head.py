"""CPU restatement of the Pair-Net head (TEST INFRASTRUCTURE, oracle/__init__.py).

`OracleCrossHead2` restates the inference half of the reference's CrossHead2:
  construction   pairnet/models/relation_heads/pairnet_head.py:24-176
  forward_head   :216-258
  forward        :260-417
  get_bboxes     :760-924
with the third-party layers taken from oracle/layers.py.  Parameter names are the
reference's so state dicts interchange with the shimmed
reference class; tests/test_oracle.py checks the two bit-for-bit on CPU.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers as L
from .matrix_learner import build_matrix_learner

INSTANCE_OFFSET = 1000  # mmdet.datasets.coco_panoptic (pairnet_head.py:16)


def _mlp3(c_in, c_mid, c_out):
    return nn.Sequential(nn.Linear(c_in, c_mid), nn.ReLU(inplace=True),
                         nn.Linear(c_mid, c_mid), nn.ReLU(inplace=True),
                         nn.Linear(c_mid, c_out))


class OracleCrossHead2(nn.Module):
    def __init__(self, num_classes, in_channels, num_relations,
                 num_obj_query=100, num_rel_query=100, mapper="conv_tiny",
                 use_mask=True, pixel_decoder=None, transformer_decoder=None,
                 feat_channels=256, out_channels=256,
                 num_transformer_feat_level=3, embed_dims=256,
                 relation_decoder=None, enforce_decoder_input_project=False,
                 n_heads=8, positional_encoding=None, **unused):
        super().__init__()
        assert not enforce_decoder_input_project
        self.num_classes, self.num_relations = num_classes, num_relations
        self.num_obj_query = self.num_queries = num_obj_query
        self.num_rel_query = num_rel_query
        self.use_mask, self.n_heads, self.embed_dims = use_mask, n_heads, embed_dims
        self.num_transformer_feat_level = num_transformer_feat_level
        self.relation_decoder = L.build_transformer_layer_sequence(relation_decoder)
        self.rel_query_embed = nn.Embedding(num_rel_query, feat_channels)
        self.rel_query_embed2 = nn.Embedding(num_rel_query * 2, feat_channels)
        self.rel_query_embed3 = nn.Embedding(num_rel_query * 2, feat_channels)
        self.rel_query_feat = nn.Embedding(num_rel_query, feat_channels)
        self.update_importance = build_matrix_learner(mapper)
        pd = dict(pixel_decoder)
        pd.update(in_channels=in_channels, feat_channels=feat_channels,
                  out_channels=out_channels)
        self.pixel_decoder = L.build_plugin_layer(pd)[1]
        self.transformer_decoder = L.build_transformer_layer_sequence(
            transformer_decoder)
        assert self.transformer_decoder.embed_dims == feat_channels
        self.decoder_positional_encoding = L.build_positional_encoding(
            positional_encoding)
        self.query_embed = nn.Embedding(num_obj_query, feat_channels)
        self.query_feat = nn.Embedding(num_obj_query, feat_channels)
        self.level_embed = nn.Embedding(num_transformer_feat_level, feat_channels)
        self.cls_embed = nn.Linear(feat_channels, num_classes + 1)
        self.mask_embed = _mlp3(feat_channels, feat_channels, out_channels)
        self.sub_query_update = _mlp3(embed_dims, embed_dims, embed_dims)
        self.obj_query_update = _mlp3(embed_dims, embed_dims, embed_dims)
        self.rel_cls_embed = nn.Linear(embed_dims, num_relations)

    # pairnet_head.py:177-193
    def init_weights(self):
        self.pixel_decoder.init_weights()
        for dec in (self.transformer_decoder, self.relation_decoder):
            for p in dec.parameters():
                if p.dim() > 1:
                    nn.init.xavier_normal_(p)

    # pairnet_head.py:216-258
    def forward_head(self, decoder_out, mask_feature, target_size):
        x = self.transformer_decoder.post_norm(decoder_out).transpose(0, 1)
        cls_pred = self.cls_embed(x)
        mask_pred = torch.einsum("bqc,bchw->bqhw", self.mask_embed(x),
                                 mask_feature)
        a = F.interpolate(mask_pred, target_size, mode="bilinear",
                          align_corners=False)
        a = a.flatten(2).unsqueeze(1).repeat((1, self.n_heads, 1, 1)).flatten(0, 1)
        return cls_pred, mask_pred, (a.sigmoid() < 0.5)

    # pairnet_head.py:260-417
    @torch.no_grad()
    def forward(self, feats, img_metas, trace=None):
        """`trace`: optional dict that receives intermediates for kernel tests."""
        bs = len(img_metas)
        mask_features, memories = self.pixel_decoder(feats)
        keys, key_pos = [], []
        for i in range(self.num_transformer_feat_level):
            m = memories[i]
            keys.append(m.flatten(2).permute(2, 0, 1)
                        + self.level_embed.weight[i].view(1, 1, -1))
            pad = m.new_zeros((bs,) + m.shape[-2:], dtype=torch.bool)
            key_pos.append(self.decoder_positional_encoding(pad)
                           .flatten(2).permute(2, 0, 1))
        q = self.query_feat.weight.unsqueeze(1).repeat((1, bs, 1))
        q_pos = self.query_embed.weight.unsqueeze(1).repeat((1, bs, 1))
        cls_pred, mask_pred, attn_mask = self.forward_head(
            q, mask_features, memories[0].shape[-2:])
        if trace is not None:
            trace.update(mask_features=mask_features, memories=memories,
                         layer_q=[], layer_attn_mask=[])
        nl = self.num_transformer_feat_level
        for i, layer in enumerate(self.transformer_decoder.layers):
            lvl = i % nl
            attn_mask[torch.where(attn_mask.sum(-1) == attn_mask.shape[-1])] = False
            if trace is not None:
                trace["layer_attn_mask"].append(attn_mask.clone())
            q = layer(query=q, key=keys[lvl], value=keys[lvl], query_pos=q_pos,
                      key_pos=key_pos[lvl], attn_masks=[attn_mask, None],
                      query_key_padding_mask=None, key_padding_mask=None)
            cls_pred, mask_pred, attn_mask = self.forward_head(
                q, mask_features, memories[(i + 1) % nl].shape[-2:])
            if trace is not None:
                trace["layer_q"].append(q.clone())
        # Pair Proposal Network; only the last layer's projection is used
        # (:322-326 project all nine and index [-1]).
        s = F.normalize(self.sub_query_update(q).transpose(0, 1), p=2, dim=-1,
                        eps=1e-12)
        o = F.normalize(self.obj_query_update(q).transpose(0, 1), p=2, dim=-1,
                        eps=1e-12)
        importance_raw = torch.matmul(s, o.transpose(1, 2))
        importance = self.update_importance(importance_raw)
        _, idx = torch.topk(importance.flatten(-2, -1), k=self.num_rel_query)
        sub_pos = torch.div(idx, self.num_obj_query, rounding_mode="trunc")
        obj_pos = torch.remainder(idx, self.num_obj_query)
        pair_feat, rel_preds = self.relation_logits(q, sub_pos, obj_pos)
        nc = cls_pred.shape[-1]
        hw = mask_pred.shape[-2:]
        g_cls = lambda p: torch.gather(cls_pred, 1, p.unsqueeze(-1).expand(-1, -1, nc))
        g_seg = lambda p: torch.gather(
            mask_pred, 1, p[..., None, None].expand(-1, -1, hw[0], hw[1]))
        if trace is not None:
            trace.update(importance_raw=importance_raw, topk_idx=idx,
                         sub_pos=sub_pos, obj_pos=obj_pos, pair_feat=pair_feat,
                         query_feat=q)
        return (dict(sub=g_cls(sub_pos), obj=g_cls(obj_pos), cls=cls_pred,
                     rel=rel_preds, importance=importance),
                dict(mask=mask_pred, sub_seg=g_seg(sub_pos),
                     obj_seg=g_seg(obj_pos)))

    # pairnet_head.py:342-378: pair features of the selected (sub, obj) queries ->
    # Relation Fusion decoder -> relation logits.  Separate so that tests can evaluate
    # it for the pair list the GPU selected when a near-tie reorders the top-k.
    @torch.no_grad()
    def relation_logits(self, q, sub_pos, obj_pos):
        bs = q.shape[1]
        ex = lambda p: p.unsqueeze(-1).repeat(1, 1, self.embed_dims).transpose(0, 1)
        pair_feat = torch.cat([torch.gather(q, 0, ex(sub_pos)),
                               torch.gather(q, 0, ex(obj_pos))], dim=0)
        r = self.rel_query_feat.weight.unsqueeze(1).repeat((1, bs, 1))
        r_pos = self.rel_query_embed.weight.unsqueeze(1).repeat((1, bs, 1))
        p_pos = self.rel_query_embed2.weight.unsqueeze(1).repeat((1, bs, 1))
        for layer in self.relation_decoder.layers:
            r = layer(query=r, key=pair_feat, value=pair_feat, query_pos=r_pos,
                      key_pos=p_pos, query_key_padding_mask=None,
                      key_padding_mask=None)
        return pair_feat, self.rel_cls_embed(r.transpose(0, 1))

    # pairnet_head.py:760-786
    def get_bboxes(self, cls_scores, mask_preds, img_metas, rescale=False):
        return [self._get_bboxes_single(
            mask_preds["mask"][i], cls_scores["cls"][i], cls_scores["sub"][i],
            cls_scores["obj"][i], cls_scores["rel"][i],
            mask_preds["sub_seg"][i], mask_preds["obj_seg"][i],
            img_metas[i]["img_shape"], img_metas[i]["scale_factor"], rescale)
            for i in range(len(img_metas))]

    # pairnet_head.py:788-924
    def _get_bboxes_single(self, all_masks, all_cls, s_cls, o_cls, r_cls,
                           s_seg, o_seg, img_shape, scale_factor, rescale=False):
        size = (round(img_shape[0] / scale_factor[1]),
                round(img_shape[1] / scale_factor[0]))
        up = lambda m: F.interpolate(m.unsqueeze(1), size=size, mode="bilinear",
                                     align_corners=False).squeeze(1)
        s_labels = F.softmax(s_cls, -1)[..., :-1].argmax(-1) + 1
        o_labels = F.softmax(o_cls, -1)[..., :-1].argmax(-1) + 1
        r_dists = F.softmax(r_cls, -1).reshape(-1, self.num_relations)
        r_dists = torch.cat([torch.zeros(self.num_rel_query, 1), r_dists], -1)
        labels = torch.cat((s_labels, o_labels), 0)
        scores, cls_ids = F.softmax(all_cls, -1)[..., :-1].max(-1)
        all_masks = up(all_masks)
        masks = torch.cat((torch.sigmoid(up(s_seg)) > 0.5,
                           torch.sigmoid(up(o_seg)) > 0.5), 0)
        # (sic) compares with the LAST REAL class id (Appendix B quirk)
        keep = (cls_ids != self.num_classes - 1) & (scores > 0.5)
        cls_ids, all_masks, scores = cls_ids[keep], all_masks[keep], scores[keep]
        h, w = all_masks.shape[-2:]
        if cls_ids.numel() == 0:
            pan_img = torch.ones(size).to(torch.long)
        else:
            flat = all_masks.flatten(1)
            stuff = {}
            for k, lab in enumerate(cls_ids.tolist()):
                if lab >= 80:
                    stuff.setdefault(lab, []).append(k)

            def ids_area(flat, n, dedup):
                m_id = flat.transpose(0, 1).softmax(-1).argmax(-1).view(h, w)
                if dedup:
                    for eq in stuff.values():
                        if len(eq) > 1:
                            for e in eq:
                                m_id.masked_fill_(m_id.eq(e), eq[0])
                seg = (m_id * INSTANCE_OFFSET + cls_ids[m_id]).view(h, w).long()
                return [int(m_id.eq(i).sum()) for i in range(n)], seg

            area, pan_img = ids_area(flat, len(scores), True)
            while True:
                small = torch.tensor([a <= 4 for a in area], dtype=torch.bool)
                if not small.any():
                    break
                scores, cls_ids, flat = scores[~small], cls_ids[~small], flat[~small]
                area, pan_img = ids_area(flat, len(scores), False)
        n2 = self.num_rel_query * 2
        return (torch.zeros((n2, 5)), labels,
                torch.arange(n2, dtype=torch.int).reshape(2, -1).T, masks,
                pan_img, torch.zeros(self.num_rel_query),
                torch.zeros(self.num_rel_query), r_dists)

    def simple_test_bboxes(self, feats, img_metas, rescale=False):
        return self.get_bboxes(*self.forward(feats, img_metas), img_metas,
                               rescale=rescale)
