"""CPU restatement of the sibling head PSGTrHead2 ("PSGTR on Mask2Former": every query is
a triplet; subject / object / predicate heads on the shared trunk).  TEST INFRASTRUCTURE.

Follows pairnet/models/relation_heads/psgtr_head2.py: construction :24-252, forward_head
:288-343, forward :345-444, get_bboxes :1063-1152.  Two reference quirks are preserved
on purpose:
  * :404-411 unpacks forward_head's fourth AND fifth outputs into `mask_pred_sub`, so the
    returned `sub_seg` is the last layer's obj_mask_embed mask and the returned `obj_seg`
    is the object mask of the INITIAL forward_head call (learned query features, no
    decoder layer); sub_mask_embed never reaches an output;
  * get_bboxes (:1065-1072) indexes the leading (stacked-layer, size 1) dimension with
    the image id, so the reference itself only works with one image per call.
Pinned bit-for-bit against the reference class run under shims (tests/test_oracle.py).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers as L


class OraclePSGTrHead2(nn.Module):
    def __init__(self, num_classes, num_relations, in_channels=(256, 512, 1024, 2048),
                 use_mask=True, num_obj_query=100, n_heads=8, embed_dims=256, pixel_decoder=None,
                 transformer_decoder=None, feat_channels=256, out_channels=256,
                 num_transformer_feat_level=3, enforce_decoder_input_project=False,
                 positional_encoding=dict(type="SinePositionalEncoding", num_feats=128,
                                          normalize=True), test_cfg=None, **unused):
        super().__init__()
        assert not enforce_decoder_input_project
        self.num_queries, self.num_classes, self.num_relations = num_obj_query, num_classes, num_relations
        self.use_mask, self.n_heads, self.test_cfg = use_mask, n_heads, test_cfg
        self.num_transformer_feat_level = num_transformer_feat_level
        pd = dict(pixel_decoder)
        pd.update(in_channels=list(in_channels), feat_channels=feat_channels,
                  out_channels=out_channels)
        self.pixel_decoder = L.build_plugin_layer(pd)[1]
        self.transformer_decoder = L.build_transformer_layer_sequence(transformer_decoder)
        self.decoder_positional_encoding = L.build_positional_encoding(positional_encoding)
        self.query_embed = nn.Embedding(num_obj_query, feat_channels)
        self.query_feat = nn.Embedding(num_obj_query, feat_channels)
        self.level_embed = nn.Embedding(num_transformer_feat_level, feat_channels)
        self.sub_cls_embed = nn.Linear(feat_channels, num_classes + 1)
        self.obj_cls_embed = nn.Linear(feat_channels, num_classes + 1)
        self.rel_cls_embed = nn.Linear(feat_channels, num_relations + 1)
        mlp3 = lambda: nn.Sequential(
            nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
            nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
            nn.Linear(feat_channels, out_channels))
        self.sub_mask_embed, self.obj_mask_embed, self.mask_embed = mlp3(), mlp3(), mlp3()

    def forward_head(self, decoder_out, mask_feature, attn_mask_target_size):   # :288-343
        x = self.transformer_decoder.post_norm(decoder_out).transpose(0, 1)
        cls_s, cls_o, cls_r = self.sub_cls_embed(x), self.obj_cls_embed(x), self.rel_cls_embed(x)
        mask_pred = torch.einsum("bqc,bchw->bqhw", self.mask_embed(x), mask_feature)
        attn = F.interpolate(mask_pred, attn_mask_target_size, mode="bilinear", align_corners=False)
        m_s = torch.einsum("bqc,bchw->bqhw", self.sub_mask_embed(x), mask_feature)
        m_o = torch.einsum("bqc,bchw->bqhw", self.obj_mask_embed(x), mask_feature)
        attn = attn.flatten(2).unsqueeze(1).repeat((1, self.n_heads, 1, 1)).flatten(0, 1)
        return cls_s, cls_o, cls_r, m_s, m_o, (attn.sigmoid() < 0.5).detach()

    @torch.no_grad()
    def forward(self, feats, img_metas):                                        # :345-444
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
        cls_s, cls_o, cls_r, m_s, m_o, attn = self.forward_head(q, mask_features,
                                                                memories[0].shape[-2:])
        nl = self.num_transformer_feat_level
        for i, layer in enumerate(self.transformer_decoder.layers):
            lvl = i % nl
            attn[torch.where(attn.sum(-1) == attn.shape[-1])] = False
            q = layer(query=q, key=keys[lvl], value=keys[lvl], query_pos=q_pos,
                      key_pos=key_pos[lvl], attn_masks=[attn, None],
                      query_key_padding_mask=None, key_padding_mask=None)
            # the reference unpacks the 4th and then the 5th element into mask_pred_sub
            # (:404-411): m_s ends as this call's obj_mask_embed mask, m_o keeps the value
            # of the initial call
            cls_s, cls_o, cls_r, _, m_s, attn = self.forward_head(
                q, mask_features, memories[(i + 1) % nl].shape[-2:])
        return (dict(sub=cls_s.unsqueeze(0), obj=cls_o.unsqueeze(0), rel=cls_r.unsqueeze(0)),
                dict(sub_seg=m_s.unsqueeze(0), obj_seg=m_o.unsqueeze(0)))

    def get_bboxes(self, cls_scores, mask_preds, img_metas, rescale=False):      # :1063-1085
        return [self._get_bboxes_single(
            cls_scores["sub"][i], cls_scores["obj"][i], cls_scores["rel"][i],
            mask_preds["sub_seg"][i], mask_preds["obj_seg"][i], img_metas[i]["img_shape"],
            img_metas[i]["scale_factor"], rescale) for i in range(len(img_metas))]

    def _get_bboxes_single(self, s_cls, o_cls, r_cls, s_seg, o_seg, img_shape, scale_factor,
                           rescale=False):                                      # :1087-1152
        size = (round(img_shape[0] / scale_factor[1]), round(img_shape[1] / scale_factor[0]))
        s_labels = F.softmax(s_cls, -1)[..., :-1].squeeze(0).argmax(-1) + 1
        o_labels = F.softmax(o_cls, -1)[..., :-1].squeeze(0).argmax(-1) + 1
        r_dists = F.softmax(r_cls, -1).squeeze(0)
        up = lambda m: torch.sigmoid(F.interpolate(m, size=size, mode="bilinear",
                                                   align_corners=False).squeeze(0)) > 0.5
        masks = torch.cat((up(s_seg), up(o_seg)), 0)
        n = self.num_queries
        return (torch.zeros((2 * n, 5)), torch.cat((s_labels, o_labels), 0),
                torch.arange(2 * n, dtype=torch.int).reshape(2, -1).T, masks,
                torch.ones(size).to(torch.long), torch.zeros(n), torch.zeros(n), r_dists)

    def simple_test_bboxes(self, feats, img_metas, rescale=False):
        return self.get_bboxes(*self.forward(feats, img_metas), img_metas, rescale=rescale)
