"""CPU restatement of the sibling head CrossHeadBaseline ("PSGFormer+": the same
Mask2Former trunk, a relation decoder that cross-attends the pixel memories, argmax
matching of relation queries to subject/object queries).  TEST INFRASTRUCTURE.

Follows pairnet/models/relation_heads/baseline.py: construction :22-194,
forward :298-443, get_bboxes :967-1160; third-party layers from oracle/layers.py.
Pinned bit-for-bit against the reference class run under shims (tests/test_oracle.py).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers as L
from .head import OracleCrossHead2

INSTANCE_OFFSET = 1000


class OracleCrossHeadBaseline(nn.Module):
    def __init__(self, num_classes, in_channels, num_relations, object_classes=None,
                 predicate_classes=None, num_obj_query=100, num_rel_query=100, use_mask=True,
                 pixel_decoder=None, transformer_decoder=None, feat_channels=256,
                 out_channels=256, num_transformer_feat_level=3, embed_dims=256,
                 relation_decoder=None, enforce_decoder_input_project=False, n_heads=8,
                 positional_encoding=None, test_cfg=dict(max_per_img=100), **unused):
        super().__init__()
        assert num_obj_query == num_rel_query and not enforce_decoder_input_project
        self.num_classes, self.num_relations = num_classes, num_relations
        self.num_obj_query = self.num_queries = num_obj_query
        self.num_rel_query, self.use_mask, self.n_heads = num_rel_query, use_mask, n_heads
        self.embed_dims, self.test_cfg = embed_dims, test_cfg
        self.num_transformer_feat_level = num_transformer_feat_level
        self.relation_decoder = L.build_transformer_layer_sequence(relation_decoder)
        self.rel_query_embed = nn.Embedding(num_rel_query, feat_channels)
        self.rel_query_feat = nn.Embedding(num_rel_query, feat_channels)
        pd = dict(pixel_decoder)
        pd.update(in_channels=in_channels, feat_channels=feat_channels, out_channels=out_channels)
        self.pixel_decoder = L.build_plugin_layer(pd)[1]
        self.transformer_decoder = L.build_transformer_layer_sequence(transformer_decoder)
        self.decoder_positional_encoding = L.build_positional_encoding(positional_encoding)
        self.query_embed = nn.Embedding(num_obj_query, feat_channels)
        self.query_feat = nn.Embedding(num_obj_query, feat_channels)
        self.level_embed = nn.Embedding(num_transformer_feat_level, feat_channels)
        self.cls_embed = nn.Linear(feat_channels, num_classes + 1)
        self.mask_embed = nn.Sequential(
            nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
            nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
            nn.Linear(feat_channels, out_channels))
        mlp2 = lambda: nn.Sequential(nn.Linear(embed_dims, embed_dims), nn.ReLU(inplace=True),
                                     nn.Linear(embed_dims, embed_dims))
        self.sub_query_update, self.obj_query_update = mlp2(), mlp2()
        self.rel_cls_embed = nn.Linear(embed_dims, num_relations + 1)

    forward_head = OracleCrossHead2.forward_head          # baseline.py:254-296, same code

    @torch.no_grad()
    def forward(self, feats, img_metas):                  # baseline.py:298-443
        bs = len(img_metas)
        mask_features, memories = self.pixel_decoder(feats)
        keys, key_pos = [], []
        for i in range(self.num_transformer_feat_level):
            m = memories[i]
            keys.append(m.flatten(2).permute(2, 0, 1) + self.level_embed.weight[i].view(1, 1, -1))
            pad = m.new_zeros((bs,) + m.shape[-2:], dtype=torch.bool)
            key_pos.append(self.decoder_positional_encoding(pad).flatten(2).permute(2, 0, 1))
        q = self.query_feat.weight.unsqueeze(1).repeat((1, bs, 1))
        q_pos = self.query_embed.weight.unsqueeze(1).repeat((1, bs, 1))
        cls_list, mask_list = [], []
        cls_pred, mask_pred, attn_mask = self.forward_head(q, mask_features, memories[0].shape[-2:])
        nl = self.num_transformer_feat_level
        for i, layer in enumerate(self.transformer_decoder.layers):
            lvl = i % nl
            attn_mask[torch.where(attn_mask.sum(-1) == attn_mask.shape[-1])] = False
            q = layer(query=q, key=keys[lvl], value=keys[lvl], query_pos=q_pos,
                      key_pos=key_pos[lvl], attn_masks=[attn_mask, None],
                      query_key_padding_mask=None, key_padding_mask=None)
            cls_pred, mask_pred, attn_mask = self.forward_head(
                q, mask_features, memories[(i + 1) % nl].shape[-2:])
            cls_list.append(cls_pred)
            mask_list.append(mask_pred)
        cls_preds, mask_preds = torch.stack(cls_list), torch.stack(mask_list)
        r = self.rel_query_feat.weight.unsqueeze(1).repeat((1, bs, 1))
        r_pos = self.rel_query_embed.weight.unsqueeze(1).repeat((1, bs, 1))
        for i, layer in enumerate(self.relation_decoder.layers):
            lvl = i % nl
            r = layer(query=r, key=keys[lvl], value=keys[lvl], query_pos=r_pos,
                      key_pos=key_pos[lvl], query_key_padding_mask=None, key_padding_mask=None)
        rel_query = r.transpose(0, 1)
        s = F.normalize(self.sub_query_update(q), p=2, dim=-1, eps=1e-12).transpose(0, 1)
        o = F.normalize(self.obj_query_update(q), p=2, dim=-1, eps=1e-12).transpose(0, 1)
        rn = F.normalize(rel_query, p=2, dim=-1, eps=1e-12)
        subject_scores = torch.matmul(rn, s.transpose(1, 2))
        object_scores = torch.matmul(rn, o.transpose(1, 2))
        sub_ids, obj_ids = subject_scores.max(-1)[1], object_scores.max(-1)[1]
        rel_preds = self.rel_cls_embed(rel_query)
        nc, hw = cls_preds.shape[-1], mask_preds.shape[-2:]
        g_cls = lambda p: torch.gather(cls_preds[-1], 1, p.unsqueeze(-1).expand(-1, -1, nc))
        g_seg = lambda p: torch.gather(mask_preds[-1], 1,
                                       p[..., None, None].expand(-1, -1, hw[0], hw[1]))
        return (dict(sub=g_cls(sub_ids), obj=g_cls(obj_ids), cls=cls_preds, rel=rel_preds,
                     subject_scores=subject_scores, object_scores=object_scores),
                dict(mask=mask_preds, sub_seg=g_seg(sub_ids), obj_seg=g_seg(obj_ids)))

    def get_bboxes(self, cls_scores, mask_preds, img_metas, rescale=False):   # :967-998
        return [self._get_bboxes_single(
            mask_preds["mask"][-1, i], cls_scores["cls"][-1, i], cls_scores["sub"][i],
            cls_scores["obj"][i], cls_scores["rel"][i], mask_preds["sub_seg"][i],
            mask_preds["obj_seg"][i], img_metas[i]["img_shape"], img_metas[i]["scale_factor"],
            rescale) for i in range(len(img_metas))]

    def _get_bboxes_single(self, all_masks, all_cls, s_cls, o_cls, r_cls, s_seg, o_seg,
                           img_shape, scale_factor, rescale=False):           # :1000-1160
        size = (round(img_shape[0] / scale_factor[1]), round(img_shape[1] / scale_factor[0]))
        k = self.test_cfg.get("max_per_img", self.num_obj_query)
        up = lambda m: F.interpolate(m.unsqueeze(1), size=size, mode="bilinear",
                                     align_corners=False).squeeze(1)
        s_scores, s_labels = F.softmax(s_cls, -1)[..., :-1].max(-1)
        o_scores, o_labels = F.softmax(o_cls, -1)[..., :-1].max(-1)
        r_lgs = F.softmax(r_cls, -1)
        r_scores, r_idx = r_lgs[..., 1:].reshape(-1).topk(k)
        r_labels = r_idx % self.num_relations + 1
        tri = torch.div(r_idx, self.num_relations, rounding_mode="trunc")
        labels = torch.cat((s_labels[tri] + 1, o_labels[tri] + 1), 0)
        r_dists = r_lgs.reshape(-1, self.num_relations + 1)[tri]
        masks = torch.cat((torch.sigmoid(up(s_seg[tri])) > 0.5,
                           torch.sigmoid(up(o_seg[tri])) > 0.5), 0)
        # the panoptic branch is the code of CrossHead2 (same source lines)
        pan_img = OracleCrossHead2._get_bboxes_single(
            self, all_masks, all_cls, s_cls, o_cls, r_cls[:, 1:], s_seg, o_seg, img_shape,
            scale_factor, rescale)[4]
        n2 = 2 * k
        return (torch.zeros((n2, 5)), labels, torch.arange(n2, dtype=torch.int).reshape(2, -1).T,
                masks, pan_img, r_scores, r_labels, r_dists)

    def simple_test_bboxes(self, feats, img_metas, rescale=False):
        return self.get_bboxes(*self.forward(feats, img_metas), img_metas, rescale=rescale)
