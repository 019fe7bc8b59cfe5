"""fp32 torch-CPU restatement of the third-party layers on the Pair-Net hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The sources of mmcv-full==1.7.0 /
mmdet==2.25.1 (README.md:80-87 of the reference) are neither under /root/reference
nor installed; the semantics below follow SURVEY.md Appendix A and the in-tree
copies the reference keeps of two of them
(pairnet/models/relation_heads/facebook_detr.py:311-353 for the attention
wrapper, :378-432 for the layer glue).

PARITY: unpinned against mmcv / mmdet themselves, but the ARITHMETIC is pinned
against an independent implementation of the same published models, HuggingFace
`transformers` (importable in this image), with identical weights through the key
converters of oracle/hf_pin.py (tests/test_oracle.py):
  * MSDeformAttnPixelDecoder (1x1 convs + GN, sine PE + level embedding, reference
    points, 6 x [MultiScaleDeformableAttention -> LN -> FFN -> LN], FPN level,
    mask_feature) == transformers Mask2FormerPixelDecoder to 2e-5 relative;
  * one masked-attention decoder layer (BaseTransformerLayer + MultiheadAttention
    wrapper + FFN, cross -> self -> ffn, post-norm) == transformers
    Mask2FormerMaskedAttentionDecoderLayer (its self-attention is not
    nn.MultiheadAttention) to 2e-5 relative;
  * the same layer in (self -> cross -> ffn) order == transformers DetrDecoderLayer
    (own attention code for both attentions) to 2e-5 relative.
The state-dict KEY NAMES of mmcv / mmdet are from memory and stay unpinned.

Every module keeps the attribute / state-dict names of the package it restates
so that a reference checkpoint would load (note N7).
Builders take the config dicts of configs/mask2former/pairnet.py:33-142.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class CfgDict(dict):
    """dict with attribute access, nested (the part of mmcv.ConfigDict the
    reference head relies on, pairnet_head.py:83-87)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for k, v in list(self.items()):
            self[k] = _wrap(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = _wrap(v)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = _wrap(v)


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, CfgDict):
        return CfgDict(v)
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


# --------------------------------------------------------------------------- #
# A1  FFN  (cfg pairnet.py:55-62, 87-95, 121-129)
# --------------------------------------------------------------------------- #
class FFN(nn.Module):
    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=None, ffn_drop=0.0, dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__()
        assert num_fcs == 2
        self.embed_dims = embed_dims
        self.add_identity = add_identity
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels),
                          nn.ReLU(inplace=True), nn.Dropout(ffn_drop)),
            nn.Linear(feedforward_channels, embed_dims),
            nn.Dropout(ffn_drop))

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return out
        if identity is None:
            identity = x
        return identity + out


# --------------------------------------------------------------------------- #
# A2  MultiheadAttention wrapper (in-tree copy: facebook_detr.py:311-353)
# --------------------------------------------------------------------------- #
class MultiheadAttention(nn.Module):
    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0,
                 dropout_layer=None, init_cfg=None, batch_first=False,
                 **kwargs):
        super().__init__()
        assert not batch_first
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop)

    def forward(self, query, key=None, value=None, identity=None,
                query_pos=None, key_pos=None, attn_mask=None,
                key_padding_mask=None, **kwargs):
        # **kwargs swallows value_pos (pairnet_head.py:373): dead input.
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None \
                and query_pos.shape == key.shape:
            key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        out = self.attn(query=query, key=key, value=value,
                        attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask,
                        need_weights=False)[0]
        return identity + out


# --------------------------------------------------------------------------- #
# A7  MultiScaleDeformableAttention (cfg pairnet.py:43-54)
# --------------------------------------------------------------------------- #
def msda_core(value, spatial_shapes, sampling_locations, attention_weights):
    """The CPU formula of mmcv's `multi_scale_deformable_attn_pytorch`.

    value (B, sumN, H, D); spatial_shapes list[(h, w)];
    sampling_locations (B, Nq, H, L, P, 2) in [0,1] (x, y);
    attention_weights (B, Nq, H, L, P).  Returns (B, Nq, H*D).
    """
    bs, _, nh, d = value.shape
    _, nq, _, nl, npnt, _ = sampling_locations.shape
    value_list = value.split([h * w for h, w in spatial_shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(spatial_shapes):
        v = value_list[lvl].flatten(2).transpose(1, 2).reshape(bs * nh, d, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode="bilinear",
                                     padding_mode="zeros",
                                     align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(bs * nh, 1, nq, nl * npnt)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1)
    return out.view(bs, nh * d, nq).transpose(1, 2).contiguous()


class MultiScaleDeformableAttention(nn.Module):
    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4,
                 im2col_step=64, dropout=0.1, batch_first=False, norm_cfg=None,
                 init_cfg=None, **kwargs):
        super().__init__()
        assert not batch_first
        self.embed_dims, self.num_heads = embed_dims, num_heads
        self.num_levels, self.num_points = num_levels, num_points
        self.batch_first = batch_first
        n = num_heads * num_levels * num_points
        self.sampling_offsets = nn.Linear(embed_dims, n * 2)
        self.attention_weights = nn.Linear(embed_dims, n)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)

    def init_weights(self):
        nn.init.constant_(self.sampling_offsets.weight, 0.0)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (
            2.0 * math.pi / self.num_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(
            self.num_heads, 1, 1, 2).repeat(1, self.num_levels,
                                            self.num_points, 1)
        for i in range(self.num_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.view(-1))
        nn.init.constant_(self.attention_weights.weight, 0.0)
        nn.init.constant_(self.attention_weights.bias, 0.0)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.constant_(self.value_proj.bias, 0.0)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.0)

    def forward(self, query, key=None, value=None, identity=None,
                query_pos=None, key_padding_mask=None, reference_points=None,
                spatial_shapes=None, level_start_index=None, **kwargs):
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        query = query.permute(1, 0, 2)
        value = value.permute(1, 0, 2)
        bs, nq, _ = query.shape
        nv = value.shape[1]
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, nv, self.num_heads, -1)
        off = self.sampling_offsets(query).view(
            bs, nq, self.num_heads, self.num_levels, self.num_points, 2)
        aw = self.attention_weights(query).view(
            bs, nq, self.num_heads, self.num_levels * self.num_points)
        aw = aw.softmax(-1).view(bs, nq, self.num_heads, self.num_levels,
                                 self.num_points)
        shapes = [(int(h), int(w)) for h, w in spatial_shapes]
        normalizer = torch.tensor([[w, h] for h, w in shapes],
                                  dtype=query.dtype)
        if reference_points.shape[-1] == 2:
            loc = reference_points[:, :, None, :, None, :] \
                + off / normalizer[None, None, None, :, None, :]
        else:   # (cx, cy, w, h) reference boxes: the decoder of a box-refining /
            #     two-stage Deformable DETR (mmcv MultiScaleDeformableAttention)
            assert reference_points.shape[-1] == 4
            loc = reference_points[:, :, None, :, None, :2] \
                + off / self.num_points \
                * reference_points[:, :, None, :, None, 2:] * 0.5
        out = msda_core(value, shapes, loc, aw)
        out = self.output_proj(out).permute(1, 0, 2)
        return out + identity


# --------------------------------------------------------------------------- #
# A3  BaseTransformerLayer (in-tree copy: facebook_detr.py:378-432)
# --------------------------------------------------------------------------- #
_ATTN = {"MultiheadAttention": MultiheadAttention,
         "MultiScaleDeformableAttention": MultiScaleDeformableAttention}


class BaseTransformerLayer(nn.Module):
    def __init__(self, attn_cfgs=None, ffn_cfgs=None, operation_order=None,
                 norm_cfg=None, init_cfg=None, batch_first=False, **kwargs):
        super().__init__()
        self.operation_order = tuple(operation_order)
        self.pre_norm = self.operation_order[0] == "norm"
        assert not self.pre_norm
        assert norm_cfg is None or dict(norm_cfg).get("type") == "LN"
        num_attn = sum(op in ("self_attn", "cross_attn")
                       for op in self.operation_order)
        self.num_attn = num_attn
        self.attentions = nn.ModuleList()
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [attn_cfgs] * num_attn
        assert len(attn_cfgs) == num_attn
        for cfg in attn_cfgs:
            cfg = dict(cfg)
            self.attentions.append(_ATTN[cfg.pop("type")](**cfg))
        self.embed_dims = self.attentions[0].embed_dims
        # mmcv's deprecated layer-level spellings (`feedforward_channels`,
        # `ffn_dropout`, `ffn_num_fcs`) are folded into ffn_cfgs
        ffn_cfgs = dict(ffn_cfgs) if ffn_cfgs is not None else dict(
            embed_dims=self.embed_dims, feedforward_channels=1024, num_fcs=2,
            ffn_drop=0.0)
        for old_name, new_name in (("feedforward_channels", "feedforward_channels"),
                                   ("ffn_dropout", "ffn_drop"),
                                   ("ffn_num_fcs", "num_fcs")):
            if old_name in kwargs:
                ffn_cfgs[new_name] = kwargs[old_name]
        ffn_cfgs.setdefault("embed_dims", self.embed_dims)
        self.ffns = nn.ModuleList()
        for _ in range(self.operation_order.count("ffn")):
            cfg = dict(ffn_cfgs)
            cfg.pop("type", None)
            self.ffns.append(FFN(**cfg))
        self.norms = nn.ModuleList(
            nn.LayerNorm(self.embed_dims)
            for _ in range(self.operation_order.count("norm")))

    def forward(self, query, key=None, value=None, query_pos=None,
                key_pos=None, attn_masks=None, query_key_padding_mask=None,
                key_padding_mask=None, **kwargs):
        ni = ai = fi = 0
        if attn_masks is None:
            attn_masks = [None] * self.num_attn
        for op in self.operation_order:
            if op == "self_attn":
                query = self.attentions[ai](
                    query, query, query, None, query_pos=query_pos,
                    key_pos=query_pos, attn_mask=attn_masks[ai],
                    key_padding_mask=query_key_padding_mask, **kwargs)
                ai += 1
            elif op == "cross_attn":
                query = self.attentions[ai](
                    query, key, value, None, query_pos=query_pos,
                    key_pos=key_pos, attn_mask=attn_masks[ai],
                    key_padding_mask=key_padding_mask, **kwargs)
                ai += 1
            elif op == "norm":
                query = self.norms[ni](query)
                ni += 1
            elif op == "ffn":
                query = self.ffns[fi](query, None)
                fi += 1
        return query


# --------------------------------------------------------------------------- #
# A4  DetrTransformerEncoder / DetrTransformerDecoder
# --------------------------------------------------------------------------- #
class TransformerLayerSequence(nn.Module):
    def __init__(self, transformerlayers=None, num_layers=None,
                 return_intermediate=False, init_cfg=None, is_decoder=False,
                 **kwargs):
        super().__init__()
        self.num_layers = num_layers
        self.layers = nn.ModuleList()
        for _ in range(num_layers):
            cfg = dict(transformerlayers)
            cfg.pop("type", None)
            self.layers.append(BaseTransformerLayer(**cfg))
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm
        # decoder: LN; encoder with post-norm layers: None (Appendix A4)
        self.post_norm = nn.LayerNorm(self.embed_dims) if is_decoder else None

    def forward(self, query, key, value, **kwargs):
        for layer in self.layers:
            query = layer(query, key, value, **kwargs)
        return query


class DetrTransformerEncoder(TransformerLayerSequence):
    def __init__(self, **cfg):
        super().__init__(is_decoder=False, **cfg)


class DetrTransformerDecoder(TransformerLayerSequence):
    """Only `.layers`, `.post_norm`, `.embed_dims` are used by the head
    (pairnet_head.py:95-96, 236, 297, 366)."""

    def __init__(self, **cfg):
        super().__init__(is_decoder=True, **cfg)


_SEQ = {"DetrTransformerEncoder": DetrTransformerEncoder,
        "DetrTransformerDecoder": DetrTransformerDecoder}


def build_transformer_layer_sequence(cfg):
    cfg = dict(cfg)
    return _SEQ[cfg.pop("type")](**cfg)


# --------------------------------------------------------------------------- #
# A5  SinePositionalEncoding (cfg pairnet.py:140-142)
# --------------------------------------------------------------------------- #
class SinePositionalEncoding(nn.Module):
    def __init__(self, num_feats, temperature=10000, normalize=False,
                 scale=2 * math.pi, eps=1e-6, offset=0.0, init_cfg=None):
        super().__init__()
        self.num_feats, self.temperature = num_feats, temperature
        self.normalize, self.scale, self.eps = normalize, scale, eps
        self.offset = offset

    def forward(self, mask):
        mask = mask.to(torch.int)
        not_mask = 1 - mask
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y_embed = (y_embed + self.offset) / \
                (y_embed[:, -1:, :] + self.eps) * self.scale
            x_embed = (x_embed + self.offset) / \
                (x_embed[:, :, -1:] + self.eps) * self.scale
        dim_t = torch.arange(self.num_feats, dtype=torch.float32)
        dim_t = self.temperature ** (2 * (dim_t // 2) / self.num_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        b, h, w = mask.size()
        pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(),
                             pos_x[:, :, :, 1::2].cos()), dim=4).view(b, h, w, -1)
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(),
                             pos_y[:, :, :, 1::2].cos()), dim=4).view(b, h, w, -1)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def build_positional_encoding(cfg):
    cfg = dict(cfg)
    assert cfg.pop("type") == "SinePositionalEncoding"
    return SinePositionalEncoding(**cfg)


# --------------------------------------------------------------------------- #
# A8  ConvModule, A6  MSDeformAttnPixelDecoder (cfg pairnet.py:33-71)
# --------------------------------------------------------------------------- #
class ConvModule(nn.Module):
    """conv -> GroupNorm -> (ReLU); names `.conv`, `.gn`."""

    def __init__(self, cin, cout, k, padding=0, bias=False, groups=32,
                 act=False):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding, bias=bias)
        self.gn = nn.GroupNorm(groups, cout)
        self.act = act

    def forward(self, x):
        x = self.gn(self.conv(x))
        return F.relu(x) if self.act else x


class MSDeformAttnPixelDecoder(nn.Module):
    def __init__(self, in_channels=(256, 512, 1024, 2048),
                 strides=(4, 8, 16, 32), feat_channels=256, out_channels=256,
                 num_outs=3, norm_cfg=None, act_cfg=None, encoder=None,
                 positional_encoding=None, init_cfg=None, **kwargs):
        super().__init__()
        assert norm_cfg["type"] == "GN"
        groups = norm_cfg["num_groups"]
        self.strides = strides
        self.num_input_levels = len(in_channels)
        self.num_encoder_levels = \
            encoder["transformerlayers"]["attn_cfgs"]["num_levels"]
        self.num_outs = num_outs
        self.input_convs = nn.ModuleList()
        for i in range(self.num_input_levels - 1,
                       self.num_input_levels - self.num_encoder_levels - 1,
                       -1):
            self.input_convs.append(ConvModule(in_channels[i], feat_channels,
                                               1, bias=True, groups=groups))
        self.encoder = build_transformer_layer_sequence(encoder)
        if positional_encoding is None:   # mmdet's default argument (msdeformattn_pixel_decoder.py)
            positional_encoding = dict(type="SinePositionalEncoding", num_feats=128,
                                       normalize=True)
        self.postional_encoding = build_positional_encoding(
            positional_encoding)  # (sic) attribute name as in mmdet
        self.level_encoding = nn.Embedding(self.num_encoder_levels,
                                           feat_channels)
        self.lateral_convs = nn.ModuleList()
        self.output_convs = nn.ModuleList()
        for i in range(self.num_input_levels - self.num_encoder_levels - 1,
                       -1, -1):
            self.lateral_convs.append(ConvModule(in_channels[i], feat_channels,
                                                 1, bias=False, groups=groups))
            self.output_convs.append(ConvModule(feat_channels, feat_channels,
                                                3, padding=1, bias=False,
                                                groups=groups, act=True))
        self.mask_feature = nn.Conv2d(feat_channels, out_channels, 1)

    def init_weights(self):
        for m in self.input_convs:
            nn.init.xavier_uniform_(m.conv.weight, gain=1)
            nn.init.constant_(m.conv.bias, 0)
        for m in list(self.lateral_convs) + list(self.output_convs):
            nn.init.kaiming_uniform_(m.conv.weight, a=1, mode="fan_in",
                                     nonlinearity="leaky_relu")
        nn.init.kaiming_uniform_(self.mask_feature.weight, a=1, mode="fan_in",
                                 nonlinearity="leaky_relu")
        nn.init.constant_(self.mask_feature.bias, 0)
        nn.init.normal_(self.level_encoding.weight, 0, 1)
        for p in self.encoder.parameters():
            if p.dim() > 1:
                nn.init.xavier_normal_(p)
        for layer in self.encoder.layers:
            for attn in layer.attentions:
                attn.init_weights()

    def forward(self, feats):
        bs = feats[0].shape[0]
        tokens, poss, shapes, refs = [], [], [], []
        for i in range(self.num_encoder_levels):
            lvl = self.num_input_levels - i - 1
            feat = feats[lvl]
            proj = self.input_convs[i](feat)
            h, w = feat.shape[-2:]
            pad = feat.new_zeros((bs, h, w), dtype=torch.bool)
            pos = self.postional_encoding(pad) \
                + self.level_encoding.weight[i].view(1, -1, 1, 1)
            ys = (torch.arange(h, dtype=torch.float32) + 0.5) \
                * self.strides[lvl]
            xs = (torch.arange(w, dtype=torch.float32) + 0.5) \
                * self.strides[lvl]
            yy, xx = torch.meshgrid(ys, xs, indexing="ij")
            ref = torch.stack([xx.reshape(-1), yy.reshape(-1)], -1)
            ref = ref / (torch.tensor([[w, h]], dtype=torch.float32)
                         * self.strides[lvl])
            tokens.append(proj.flatten(2).permute(2, 0, 1))
            poss.append(pos.flatten(2).permute(2, 0, 1))
            shapes.append((h, w))
            refs.append(ref)
        tokens = torch.cat(tokens, 0)
        poss = torch.cat(poss, 0)
        starts = [0]
        for h, w in shapes[:-1]:
            starts.append(starts[-1] + h * w)
        ref = torch.cat(refs, 0)[None, :, None].repeat(
            bs, 1, self.num_encoder_levels, 1)
        memory = self.encoder(
            query=tokens, key=None, value=None, query_pos=poss, key_pos=None,
            attn_masks=None, key_padding_mask=None,
            query_key_padding_mask=None, spatial_shapes=shapes,
            reference_points=ref, level_start_index=starts)
        memory = memory.permute(1, 2, 0)
        outs = torch.split(memory, [h * w for h, w in shapes], dim=-1)
        outs = [x.reshape(bs, -1, shapes[i][0], shapes[i][1])
                for i, x in enumerate(outs)]
        for i in range(self.num_input_levels - self.num_encoder_levels - 1,
                       -1, -1):
            cur = self.lateral_convs[i](feats[i])
            y = cur + F.interpolate(outs[-1], size=cur.shape[-2:],
                                    mode="bilinear", align_corners=False)
            outs.append(self.output_convs[i](y))
        return self.mask_feature(outs[-1]), outs[:self.num_outs]


def build_plugin_layer(cfg):
    cfg = dict(cfg)
    assert cfg.pop("type") == "MSDeformAttnPixelDecoder"
    return "pixel_decoder", MSDeformAttnPixelDecoder(**cfg)
