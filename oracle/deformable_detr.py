"""CPU restatement of the Deformable-DETR trunk under CrossHeadBBox (TEST INFRASTRUCTURE).

The reference's `CrossHeadBBox` (pairnet/models/relation_heads/pairnet_bbox_head.py:22)
builds its trunk from mmdet with `build_transformer(transformer)` (:66) and its neck from
`ChannelMapper` (configs/deformable_detr/cross_r101_vg.py:20-29, 41-80).  mmdet 2.25.1 /
mmcv-full 1.7.0 are not under /root/reference and not installed, so -- as for the pixel
decoder in oracle/layers.py -- the published algorithm (Zhu et al., "Deformable DETR",
ICLR 2021, sections 4 and A.4: multi-scale deformable encoder, iterative box refinement,
two-stage proposals) is restated here under mmdet's module / parameter names, and
tests/test_oracle.py pins it to HuggingFace `transformers`' independent implementation
(DeformableDetrForObjectDetection with two_stage / with_box_refine) through
oracle/hf_pin.py::deformable_detr_to_hf.

  ChannelMapper               neck: per-level 1x1 conv + GN(32), extra 3x3 stride-2 levels
  DeformableDetrTransformer   level embeddings, valid ratios, reference points, encoder,
                              two-stage proposal generation + top-k, decoder
  DeformableDetrTransformerDecoder   per-layer reference scaling + box refinement

Call sites in the reference: pairnet_bbox_head.py:215-228 (arguments, return tuple).
"""
import math

import torch
import torch.nn as nn

from . import layers as L


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


class ChannelMapper(nn.Module):
    """mmdet ChannelMapper with norm_cfg=GN, act_cfg=None: `convs.i.{conv,gn}` on input i,
    `extra_convs.j.{conv,gn}` (3x3, stride 2) on the last input, then on the last output.
    A ConvModule followed by a norm has no conv bias."""

    def __init__(self, in_channels, out_channels, kernel_size=1, num_outs=None,
                 norm_cfg=None, act_cfg=None, conv_cfg=None, init_cfg=None, **unused):
        super().__init__()
        assert act_cfg is None and norm_cfg is not None and norm_cfg["type"] == "GN"
        groups = norm_cfg["num_groups"]
        self.convs = nn.ModuleList(
            L.ConvModule(c, out_channels, kernel_size, padding=(kernel_size - 1) // 2,
                         bias=False, groups=groups) for c in in_channels)
        num_outs = len(in_channels) if num_outs is None else num_outs
        self.extra_convs = nn.ModuleList()
        for i in range(len(in_channels), num_outs):
            cin = in_channels[-1] if i == len(in_channels) else out_channels
            m = L.ConvModule(cin, out_channels, 3, padding=1, bias=False, groups=groups)
            m.conv.stride = (2, 2)
            self.extra_convs.append(m)

    def forward(self, inputs):
        assert len(inputs) == len(self.convs)
        outs = [c(x) for c, x in zip(self.convs, inputs)]
        for i, m in enumerate(self.extra_convs):
            outs.append(m(inputs[-1] if i == 0 else outs[-1]))
        return tuple(outs)


class DeformableDetrTransformerDecoder(L.TransformerLayerSequence):
    """return_intermediate decoder: every layer sees its reference boxes scaled by the
    per-level valid ratios; with `reg_branches` the boxes are refined after each layer
    (detached), and the post-refinement boxes are what is returned per layer."""

    def __init__(self, return_intermediate=False, **cfg):
        super().__init__(is_decoder=False, **cfg)
        self.return_intermediate = return_intermediate

    def forward(self, query, *args, reference_points=None, valid_ratios=None,
                reg_branches=None, **kwargs):
        output = query
        inter, inter_refs = [], []
        for lid, layer in enumerate(self.layers):
            if reference_points.shape[-1] == 4:
                ref_in = reference_points[:, :, None] * torch.cat(
                    [valid_ratios, valid_ratios], -1)[:, None]
            else:
                ref_in = reference_points[:, :, None] * valid_ratios[:, None]
            output = layer(output, *args, reference_points=ref_in, **kwargs)
            output = output.permute(1, 0, 2)
            if reg_branches is not None:
                tmp = reg_branches[lid](output)
                if reference_points.shape[-1] == 4:
                    new_ref = (tmp + inverse_sigmoid(reference_points)).sigmoid()
                else:
                    new_ref = tmp
                    new_ref[..., :2] = tmp[..., :2] + inverse_sigmoid(reference_points)
                    new_ref = new_ref.sigmoid()
                reference_points = new_ref.detach()
            output = output.permute(1, 0, 2)
            if self.return_intermediate:
                inter.append(output)
                inter_refs.append(reference_points)
        if self.return_intermediate:
            return torch.stack(inter), torch.stack(inter_refs)
        return output, reference_points


class DeformableDetrTransformer(nn.Module):
    def __init__(self, encoder=None, decoder=None, as_two_stage=False,
                 num_feature_levels=4, two_stage_num_proposals=300, init_cfg=None,
                 **unused):
        super().__init__()
        enc = dict(encoder)
        assert enc.pop("type") == "DetrTransformerEncoder"
        self.encoder = L.DetrTransformerEncoder(**enc)
        dec = dict(decoder)
        assert dec.pop("type") == "DeformableDetrTransformerDecoder"
        self.decoder = DeformableDetrTransformerDecoder(**dec)
        self.embed_dims = self.encoder.embed_dims
        self.as_two_stage = as_two_stage
        self.num_feature_levels = num_feature_levels
        self.two_stage_num_proposals = two_stage_num_proposals
        c = self.embed_dims
        self.level_embeds = nn.Parameter(torch.zeros(num_feature_levels, c))
        if as_two_stage:
            self.enc_output = nn.Linear(c, c)
            self.enc_output_norm = nn.LayerNorm(c)
            self.pos_trans = nn.Linear(2 * c, 2 * c)
            self.pos_trans_norm = nn.LayerNorm(2 * c)
        else:
            self.reference_points = nn.Linear(c, 2)

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, L.MultiScaleDeformableAttention):
                m.init_weights()
        nn.init.normal_(self.level_embeds)

    # ---- two-stage: one proposal per encoder token (paper A.4) ----
    def gen_encoder_output_proposals(self, memory, padding_mask, spatial_shapes):
        n = memory.shape[0]
        proposals, cur = [], 0
        for lvl, (h, w) in enumerate(spatial_shapes):
            m = padding_mask[:, cur:cur + h * w].view(n, h, w, 1)
            valid_h = torch.sum(~m[:, :, 0, 0], 1)
            valid_w = torch.sum(~m[:, 0, :, 0], 1)
            gy, gx = torch.meshgrid(torch.linspace(0, h - 1, h, dtype=torch.float32),
                                    torch.linspace(0, w - 1, w, dtype=torch.float32),
                                    indexing="ij")
            grid = torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1)], -1)
            scale = torch.cat([valid_w.unsqueeze(-1), valid_h.unsqueeze(-1)], 1).view(n, 1, 1, 2)
            grid = (grid.unsqueeze(0).expand(n, -1, -1, -1) + 0.5) / scale
            wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
            proposals.append(torch.cat((grid, wh), -1).view(n, -1, 4))
            cur += h * w
        prop = torch.cat(proposals, 1)
        valid = ((prop > 0.01) & (prop < 0.99)).all(-1, keepdim=True)
        prop = torch.log(prop / (1 - prop))
        prop = prop.masked_fill(padding_mask.unsqueeze(-1), float("inf"))
        prop = prop.masked_fill(~valid, float("inf"))
        out = memory.masked_fill(padding_mask.unsqueeze(-1), 0.0)
        out = out.masked_fill(~valid, 0.0)
        return self.enc_output_norm(self.enc_output(out)), prop

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios):
        refs = []
        for lvl, (h, w) in enumerate(spatial_shapes):
            ry, rx = torch.meshgrid(torch.linspace(0.5, h - 0.5, h, dtype=torch.float32),
                                    torch.linspace(0.5, w - 0.5, w, dtype=torch.float32),
                                    indexing="ij")
            ry = ry.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * h)
            rx = rx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * w)
            refs.append(torch.stack((rx, ry), -1))
        refs = torch.cat(refs, 1)
        return refs[:, :, None] * valid_ratios[:, None]

    @staticmethod
    def get_valid_ratio(mask):
        _, h, w = mask.shape
        valid_h = torch.sum(~mask[:, :, 0], 1)
        valid_w = torch.sum(~mask[:, 0, :], 1)
        return torch.stack([valid_w.float() / w, valid_h.float() / h], -1)

    @staticmethod
    def get_proposal_pos_embed(proposals, num_pos_feats=128, temperature=10000):
        scale = 2 * math.pi
        dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
        dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
        proposals = proposals.sigmoid() * scale
        pos = proposals[:, :, :, None] / dim_t
        return torch.stack((pos[:, :, :, 0::2].sin(), pos[:, :, :, 1::2].cos()),
                           dim=4).flatten(2)

    def forward(self, mlvl_feats, mlvl_masks, query_embed, mlvl_pos_embeds,
                reg_branches=None, cls_branches=None, trace=None, **kwargs):
        assert self.as_two_stage or query_embed is not None
        feats, masks, poss, shapes = [], [], [], []
        for lvl, (feat, mask, pos) in enumerate(zip(mlvl_feats, mlvl_masks, mlvl_pos_embeds)):
            shapes.append(tuple(feat.shape[-2:]))
            feats.append(feat.flatten(2).transpose(1, 2))
            masks.append(mask.flatten(1))
            poss.append(pos.flatten(2).transpose(1, 2) + self.level_embeds[lvl].view(1, 1, -1))
        feat_f, mask_f, pos_f = torch.cat(feats, 1), torch.cat(masks, 1), torch.cat(poss, 1)
        sizes = torch.tensor([h * w for h, w in shapes])
        level_start_index = torch.cat((sizes.new_zeros(1), sizes.cumsum(0)[:-1]))
        valid_ratios = torch.stack([self.get_valid_ratio(m) for m in mlvl_masks], 1)
        ref = self.get_reference_points(shapes, valid_ratios)
        memory = self.encoder(
            query=feat_f.permute(1, 0, 2), key=None, value=None,
            query_pos=pos_f.permute(1, 0, 2), query_key_padding_mask=mask_f,
            spatial_shapes=shapes, reference_points=ref,
            level_start_index=level_start_index, valid_ratios=valid_ratios)
        memory = memory.permute(1, 0, 2)
        bs, _, c = memory.shape
        if trace is not None:
            trace.update(memory=memory, spatial_shapes=shapes)
        if self.as_two_stage:
            out_mem, out_prop = self.gen_encoder_output_proposals(memory, mask_f, shapes)
            nl = self.decoder.num_layers
            enc_cls = cls_branches[nl](out_mem)
            enc_coord_unact = reg_branches[nl](out_mem) + out_prop
            topk = torch.topk(enc_cls[..., 0], self.two_stage_num_proposals, dim=1)[1]
            topk_unact = torch.gather(enc_coord_unact, 1,
                                      topk.unsqueeze(-1).repeat(1, 1, 4)).detach()
            reference_points = topk_unact.sigmoid()
            init_ref = reference_points
            pos_trans_out = self.pos_trans_norm(
                self.pos_trans(self.get_proposal_pos_embed(topk_unact)))
            query_pos, query = torch.split(pos_trans_out, c, dim=2)
            if trace is not None:
                trace.update(enc_cls0=enc_cls[..., 0], topk_proposals=topk,
                             query=query, query_pos=query_pos)
        else:
            query_pos, query = torch.split(query_embed, c, dim=1)
            query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1)
            query = query.unsqueeze(0).expand(bs, -1, -1)
            reference_points = self.reference_points(query_pos).sigmoid()
            init_ref = reference_points
            enc_cls = enc_coord_unact = None
        inter, inter_refs = self.decoder(
            query=query.permute(1, 0, 2), key=None, value=memory.permute(1, 0, 2),
            query_pos=query_pos.permute(1, 0, 2), key_padding_mask=mask_f,
            reference_points=reference_points, spatial_shapes=shapes,
            level_start_index=level_start_index, valid_ratios=valid_ratios,
            reg_branches=reg_branches)
        return inter, init_ref, inter_refs, enc_cls, enc_coord_unact


def build_transformer(cfg):
    cfg = dict(cfg)
    assert cfg.pop("type") == "DeformableDetrTransformer"
    return DeformableDetrTransformer(**cfg)
