"""CrossHeadBBox on MI355X: the reference's box-trunk sibling head behind its own interface.

Mirrors pairnet/models/relation_heads/pairnet_bbox_head.py (`CrossHeadBBox`): constructor
keywords (:23-49), `forward(mlvl_feats, img_metas)` (:193-359), `get_bboxes` (:1012-1041),
`_get_bboxes_single` (:1043-1101), `simple_test_bboxes` (:1103-1107) and the state-dict key
names, for the configurations that are consistent with the class: a two-stage,
box-refining Deformable-DETR trunk (mmdet `DeformableDetrTransformer`, built at :66 from
configs/deformable_detr/cross_r101_vg.py:41-80) under the Pair Proposal Network and a
post-norm ReLU relation decoder (:81-117).  The PPN / Matrix Learner / top-k / relation
decoder are `CrossHead2`'s (head.py); the trunk runs on the same GEMM, deformable-attention
and attention kernels plus the glue kernels of csrc/detr.hip.  No CPU path.

Not built (they are not what the shipped configs that construct this class run):
  * the one-stage form (`as_two_stage=False`: learned `query_embedding`, 2-d reference
    points) and the form without box refinement;
  * `configs/deformable_detr/pairnet_r101_vg.py`'s pre-norm RMSNorm / SwiGLU relation decoder:
    with mmcv-full 1.7.0's FFN the SwiGLU activation halves the hidden width in front of a
    Linear that expects the full one (that config does not run on the pinned mmcv), and
    `cross_r50_coco.py` / `cross_r50_oiv6.py` pass `object_transformer=`, a keyword the
    class drops into **kwargs, leaving `transformer=None`.

Reference behaviours kept on purpose (oracle/bbox_head.py lists them): the softmax over the
QUERY axis that ranks the 300 decoder queries, the literal 100 in `idx // 100`, the dead
`rel_value_pos_embed`, the unused `relation_decoder.post_norm`.

Device data layout (fp32): encoder tokens [B, N0+N1+N2+N3, 256] with levels high -> low
resolution (the order of the neck's outputs), decoder queries [B*300, 256], kept queries
[B*100, 256].  A padded batch (`img_shape` inside `batch_input_shape`, pairnet_bbox_head.py:
196-213) takes the general path: per-image positional tables / proposal validity / valid
ratios, zeroed value rows, explicit sampling operands (`pn_token_sampling_f32` +
`pn_msda_loc_f32`) instead of the fused sampler.
"""
import math
from collections import OrderedDict

import torch

from . import hip
from . import plans
from .plans import Arena, PlanCache, measure_bytes
from .config import ConfigDict
from .head import CrossHead2, _decoder_param_shapes


class CrossHeadBBox(CrossHead2):
    """Drop-in for the reference's `CrossHeadBBox` (inference half)."""

    KEPT = 100          # the literal 100 of pairnet_bbox_head.py:253, :278-279

    def __init__(self, num_classes, num_relations, use_mask=False, num_obj_query=100,
                 num_rel_query=100, transformer=None, sync_cls_avg_factor=True, embed_dims=256,
                 relation_decoder=None, num_reg_fcs=2, as_two_stage=False,
                 with_box_refine=False,
                 positional_encoding=dict(type="SinePositionalEncoding", num_feats=128,
                                          normalize=True),
                 rel_cls_loss=None, subobj_cls_loss=None, importance_match_loss=None,
                 loss_cls=None, loss_bbox=None, loss_iou=None, train_cfg=None,
                 test_cfg=dict(max_per_img=100), init_cfg=None, **kwargs):
        if transformer is None:
            raise ValueError("CrossHeadBBox needs `transformer=` (the reference class builds "
                             "its trunk from it, pairnet_bbox_head.py:66)")
        if train_cfg:
            raise NotImplementedError("inference path only (SURVEY.md section 8)")
        transformer = ConfigDict(transformer)
        relation_decoder = ConfigDict(relation_decoder)
        if not (as_two_stage and with_box_refine and transformer.get("as_two_stage", False)):
            raise NotImplementedError("two-stage, box-refining trunk only (see module docstring)")
        enc, dec = transformer.encoder, transformer.decoder
        got = tuple(relation_decoder.transformerlayers.get("operation_order", ()))
        if got != self.RELATION_ORDER or relation_decoder.transformerlayers.get("norm_cfg"):
            raise NotImplementedError("relation decoder: post-norm LayerNorm layers in the order "
                                      "%s (got %s)" % (self.RELATION_ORDER, got))
        act = relation_decoder.transformerlayers.ffn_cfgs.get("act_cfg", dict(type="ReLU"))
        if act.get("type") != "ReLU":
            raise NotImplementedError("relation decoder FFN activation %s" % act.get("type"))
        if tuple(enc.transformerlayers.operation_order) != ("self_attn", "norm", "ffn", "norm") or \
                tuple(dec.transformerlayers.operation_order) != (
                    "self_attn", "norm", "cross_attn", "norm", "ffn", "norm"):
            raise NotImplementedError("Deformable-DETR operation orders only")
        if embed_dims != 256 or num_reg_fcs != 2 or positional_encoding["num_feats"] != 128 \
                or not positional_encoding.get("normalize", False):
            raise NotImplementedError("256 channels, 2 hidden box-branch layers, normalised "
                                      "128-feature sine encoding")
        self.pe_offset = float(positional_encoding.get("offset", 0.0))
        self.pe_temperature = float(positional_encoding.get("temperature", 10000))
        loss_cls = loss_cls or {}
        self.cls_out_channels = num_classes if loss_cls.get("use_sigmoid", False) \
            else num_classes + 1
        self.num_classes, self.num_relations = num_classes, num_relations
        self.num_queries = num_obj_query                   # (the reference's attribute)
        self.num_proposals = transformer.get("two_stage_num_proposals", 300)
        self.num_obj_query = self.KEPT                     # the PPN's query count
        self.num_rel_query = num_rel_query
        self.use_mask = use_mask
        self.embed_dims, self.n_heads, self.num_heads = embed_dims, 8, 8
        self.as_two_stage, self.with_box_refine = True, True
        self.num_levels = transformer.get("num_feature_levels", 4)
        self.num_enc_layers, self.num_dec_layers = enc.num_layers, dec.num_layers
        self.enc_ffn = enc.transformerlayers.get("feedforward_channels", 1024)
        self.dec_ffn = dec.transformerlayers.get("feedforward_channels", 1024)
        self.num_rel_layers = relation_decoder.num_layers
        self.rel_ffn = relation_decoder.transformerlayers.ffn_cfgs.feedforward_channels
        self.test_cfg, self.train_cfg = test_cfg, None
        if self.num_levels != 4 or self.num_proposals > 512 or self.num_proposals < self.KEPT \
                or self.cls_out_channels > 256 or num_rel_query > 128:
            raise NotImplementedError("4 levels, <= 512 proposals, <= 256 classes")
        self._params = OrderedDict((k, torch.zeros(s)) for k, s in self.param_shapes().items())
        self.device, self.w = None, None
        self._plans, self._post, self._consts = PlanCache(), OrderedDict(), {}
        self._arenas, self._post_arenas, self._pe, self._box = {}, {}, OrderedDict(), OrderedDict()
        self._pan_jobs = []
        self.use_graphs = False
        self.grid_reserve = 0
        self.fuse_ppn_front = True
        self.enc_fused_ln = ("proj", "ffn")     # see head.py
        self.init_weights()

    # ------------------------------------------------------------------ params
    def param_shapes(self):
        """The reference's state-dict names in registration order
        (pairnet_bbox_head.py:59-66, :103-155; mmdet DeformableDetrTransformer)."""
        s = OrderedDict()
        R, nc = self.num_rel_query, self.cls_out_channels
        _decoder_param_shapes("relation_decoder", self.num_rel_layers, self.rel_ffn, s)
        s["rel_query_pos_embed.weight"] = (R, 256)
        s["rel_key_pos_embed.weight"] = (2 * R, 256)
        s["rel_value_pos_embed.weight"] = (2 * R, 256)     # dead weight (see oracle/bbox_head.py)
        s["rel_query_feat.weight"] = (R, 256)
        for i, (ci, co) in enumerate(((1, 64), (64, 64), (64, 1))):
            s["update_importance.conv_layers.%d.0.weight" % i] = (co, ci, 7, 7)
            s["update_importance.conv_layers.%d.0.bias" % i] = (co,)
        t = "transformer."
        s[t + "level_embeds"] = (4, 256)

        def msda(p):
            for name, n in (("sampling_offsets", 256), ("attention_weights", 128),
                            ("value_proj", 256), ("output_proj", 256)):
                s[p + name + ".weight"] = (n, 256)
                s[p + name + ".bias"] = (n,)

        def ffn_norms(p, ffn, norms):
            s[p + "ffns.0.layers.0.0.weight"] = (ffn, 256)
            s[p + "ffns.0.layers.0.0.bias"] = (ffn,)
            s[p + "ffns.0.layers.1.weight"] = (256, ffn)
            s[p + "ffns.0.layers.1.bias"] = (256,)
            for n in range(norms):
                s[p + "norms.%d.weight" % n] = (256,)
                s[p + "norms.%d.bias" % n] = (256,)

        for i in range(self.num_enc_layers):
            p = t + "encoder.layers.%d." % i
            msda(p + "attentions.0.")
            ffn_norms(p, self.enc_ffn, 2)
        for i in range(self.num_dec_layers):
            p = t + "decoder.layers.%d." % i
            s[p + "attentions.0.attn.in_proj_weight"] = (768, 256)
            s[p + "attentions.0.attn.in_proj_bias"] = (768,)
            s[p + "attentions.0.attn.out_proj.weight"] = (256, 256)
            s[p + "attentions.0.attn.out_proj.bias"] = (256,)
            msda(p + "attentions.1.")
            ffn_norms(p, self.dec_ffn, 3)
        for name, n in (("enc_output", 256), ("pos_trans", 512)):
            s[t + name + ".weight"] = (n, n)
            s[t + name + ".bias"] = (n,)
            s[t + name + "_norm.weight"] = (n,)
            s[t + name + "_norm.bias"] = (n,)
        for mlp in ("sub_query_update", "obj_query_update"):
            for j in (0, 2, 4):
                s["%s.%d.weight" % (mlp, j)] = (256, 256)
                s["%s.%d.bias" % (mlp, j)] = (256,)
        s["rel_cls_embed.weight"] = (self.num_relations, 256)
        s["rel_cls_embed.bias"] = (self.num_relations,)
        npred = self.num_dec_layers + 1
        for i in range(npred):
            s["cls_branches.%d.weight" % i] = (nc, 256)
            s["cls_branches.%d.bias" % i] = (nc,)
        for i in range(npred):
            for j, n in ((0, 256), (2, 256), (4, 4)):
                s["reg_branches.%d.%d.weight" % (i, j)] = (n, 256)
                s["reg_branches.%d.%d.bias" % (i, j)] = (n,)
        return s

    def init_weights(self, seed=0):
        """Random init in the families the reference ends up with (pairnet_bbox_head.py:157-171
        + mmdet DeformableDetrTransformer.init_weights): not its RNG stream."""
        g = torch.Generator().manual_seed(seed)
        U = lambda shape, b: (torch.rand(shape, generator=g) * 2 - 1) * b
        N = lambda shape, std: torch.randn(shape, generator=g) * std
        for k, p in self._params.items():
            shape, leaf = tuple(p.shape), k.rsplit(".", 1)[-1]
            if ".norms." in k or "_norm." in k:
                v = torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
            elif k == "transformer.level_embeds" or k.endswith("_embed.weight") \
                    or k == "rel_query_feat.weight":
                v = N(shape, 1.0)
            elif len(shape) == 1:
                v = torch.zeros(shape)
            elif k.startswith("relation_decoder."):
                v = N(shape, math.sqrt(2.0 / (shape[0] + shape[1])))
            else:
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
                v = U(shape, math.sqrt(6.0 / (fan_in + shape[0])))
            p.copy_(v)
        thetas = torch.arange(8, dtype=torch.float32) * (2.0 * math.pi / 8)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(8, 1, 1, 2).repeat(1, 4, 4, 1)
        for i in range(4):
            grid[:, :, i, :] *= i + 1
        for k in self._params:
            if k.endswith("sampling_offsets.weight") or k.endswith("attention_weights.weight") \
                    or k.endswith("attention_weights.bias"):
                self._params[k].zero_()
            elif k.endswith("sampling_offsets.bias"):
                self._params[k].copy_(grid.reshape(-1))
        bias_init = -math.log((1 - 0.01) / 0.01)        # bias_init_with_prob(0.01)
        for i in range(self.num_dec_layers + 1):
            self._params["cls_branches.%d.bias" % i].fill_(bias_init)
            self._params["reg_branches.%d.4.weight" % i].zero_()
        self._params["reg_branches.0.4.bias"][2:] = -2.0
        self.w, self._plans, self._consts, self._box = None, PlanCache(), {}, OrderedDict()

    # ----------------------------------------------------------------- packing
    def _pack(self):
        if self.device is None or self.device.type != "cuda":
            raise RuntimeError("CrossHeadBBox runs on an MI355X only: call .to('cuda:0'); "
                               "there is no CPU path")
        hip.lib()
        w = {k: v.to(self.device).contiguous() for k, v in self._params.items()}
        t = "transformer."
        for i in range(self.num_enc_layers):
            p = t + "encoder.layers.%d.attentions.0." % i
            w[p + "voa.weight"] = torch.cat([w[p + "value_proj.weight"],
                                             w[p + "sampling_offsets.weight"],
                                             w[p + "attention_weights.weight"]], 0).contiguous()
            w[p + "voa.bias"] = torch.cat([w[p + "value_proj.bias"], w[p + "sampling_offsets.bias"],
                                           w[p + "attention_weights.bias"]], 0).contiguous()
        for i in range(self.num_dec_layers):
            p = t + "decoder.layers.%d." % i
            self._pack_vqk(w, p + "attentions.0.attn.")
            a = p + "attentions.1."
            w[a + "oa.weight"] = torch.cat([w[a + "sampling_offsets.weight"],
                                            w[a + "attention_weights.weight"]], 0).contiguous()
            w[a + "oa.bias"] = torch.cat([w[a + "sampling_offsets.bias"],
                                          w[a + "attention_weights.bias"]], 0).contiguous()
        # CrossHead2's relation stage under its own parameter names
        w["rel_query_embed.weight"] = w["rel_query_pos_embed.weight"]
        w["rel_query_embed2.weight"] = w["rel_key_pos_embed.weight"]
        for i in range(self.num_rel_layers):
            self._pack_vqk(w, "relation_decoder.layers.%d.attentions.1.attn." % i)
            a = "relation_decoder.layers.%d.attentions.0.attn." % i
            W, b = w[a + "in_proj_weight"], w[a + "in_proj_bias"]
            w[a + "vk.weight"] = torch.cat([W[512:], W[256:512]], 0).contiguous()
            w[a + "vk.bias"] = torch.cat([b[512:], b[256:512]], 0).contiguous()
        ml = "update_importance.conv_layers."
        w[ml + "0.0.weight"] = w[ml + "0.0.weight"].reshape(64, 49).contiguous()
        w[ml + "1.0.weight"] = w[ml + "1.0.weight"].permute(0, 2, 3, 1).reshape(64, -1).contiguous()
        w[ml + "2.0.weight"] = w[ml + "2.0.weight"].reshape(64, 49).t().contiguous()
        self.w = w

    @staticmethod
    def proposals(shapes, valid_hw=None):
        """`gen_encoder_output_proposals` of one image (mmdet DeformableDetrTransformer; restated
        in oracle/deformable_detr.py): one box per token, (x + .5) / valid_W, (y + .5) / valid_H,
        side 0.05 * 2^level, as logits; +inf and valid = 0 where a coordinate leaves
        (0.01, 0.99) or the token is padding.  `valid_hw`: per level (valid_h, valid_w) of a
        padded image (default: the whole maps).  Input-independent: computed once per shape on
        the host."""
        out, pad = [], []
        for lvl, (h, w) in enumerate(shapes):
            vh, vw = valid_hw[lvl] if valid_hw is not None else (h, w)
            gy, gx = torch.meshgrid(torch.linspace(0, h - 1, h, dtype=torch.float32),
                                    torch.linspace(0, w - 1, w, dtype=torch.float32),
                                    indexing="ij")
            grid = torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1)], -1)
            grid = (grid + 0.5) / torch.tensor([vw, vh]).view(1, 1, 2)
            wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
            out.append(torch.cat((grid, wh), -1).view(-1, 4))
            pad.append(((gy >= vh) | (gx >= vw)).view(-1))
        prop, pad = torch.cat(out, 0), torch.cat(pad, 0)
        valid = ((prop > 0.01) & (prop < 0.99)).all(-1) & ~pad
        prop = torch.log(prop / (1 - prop))
        prop = prop.masked_fill(~valid.unsqueeze(-1), float("inf"))
        return prop, valid

    @staticmethod
    def level_valid_sizes(img_metas, shapes):
        """Per image and level (valid_h, valid_w): pairnet_bbox_head.py:196-213 -- the image
        mask is nearest-resampled to every level, and the trunk counts the unmasked rows /
        columns of its first column / row."""
        import torch.nn.functional as F
        ih, iw = img_metas[0]["batch_input_shape"]
        out = []
        for m in img_metas:
            h, w = m["img_shape"][:2]
            mask = torch.ones(1, 1, ih, iw)
            mask[..., :h, :w] = 0
            per = []
            for (fh, fw) in shapes:
                ml = F.interpolate(mask, size=(fh, fw)).to(torch.bool)[0, 0]
                per.append((int((~ml[:, 0]).sum()), int((~ml[0, :]).sum())))
            out.append(tuple(per))
        return tuple(out)

    BOX_SHAPES = 16      # per-(shapes, padding) constant sets kept (LRU)

    def _drop_weight_state(self):
        super()._drop_weight_state()
        self._box = OrderedDict()      # (the position tables carry the level embeddings)

    def _box_constants(self, B, shapes, valid_sizes):
        """What depends on the level shapes (and, for a padded batch, on every image's valid
        sizes) but not on the input: position tables with the level embeddings, the encoder's
        proposal boxes and their validity, token validity and valid ratios, the level geometry
        as device tensors.  Read-only, shared by the slots, one set per (batch, shapes,
        padding) in a small LRU; `ready` is an event behind the kernels / copies that fill it."""
        key = (B, tuple(shapes), valid_sizes)
        c = self._box.get(key)
        if c is not None:
            self._box.move_to_end(key)
            return c
        dev, w = self.device, self.w
        E = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        N = [h * wd for h, wd in shapes]
        start = [sum(N[:l]) for l in range(4)]
        SN = sum(N)
        c = dict(padded=valid_sizes is not None)
        if not c["padded"]:
            c["enc_pos"] = E(SN, 256)
            for l, (h, wd) in enumerate(shapes):
                hip.sine_pe(c["enc_pos"][start[l]:start[l] + N[l]],
                            w["transformer.level_embeds"][l], h, wd,
                            temperature=self.pe_temperature, offset=self.pe_offset)
            prop, valid = self.proposals(shapes)
            c["prop"] = prop.unsqueeze(0).expand(B, SN, 4).contiguous().to(dev)
            c["valid"] = valid.to(torch.uint8).to(dev)
            c["vr"] = c["tok_valid"] = None
        else:
            # padded batch: per-image positional tables, validity and valid ratios
            c["enc_pos"] = E(B, SN, 256)
            props, valids, toks, vrs = [], [], [], []
            for b in range(B):
                for l, (h, wd) in enumerate(shapes):
                    hip.sine_pe(c["enc_pos"][b, start[l]:start[l] + N[l]],
                                w["transformer.level_embeds"][l], h, wd,
                                temperature=self.pe_temperature, offset=self.pe_offset,
                                valid=valid_sizes[b][l])
                prop, valid = self.proposals(shapes, valid_sizes[b])
                props.append(prop)
                valids.append(valid)
                toks.append(torch.cat([
                    ((torch.arange(h).view(-1, 1) < vh) & (torch.arange(wd).view(1, -1) < vw)).view(-1)
                    for (h, wd), (vh, vw) in zip(shapes, valid_sizes[b])]))
                vrs.append(torch.stack([      # `valid_W.float() / W`: float32 quotients
                    torch.stack([torch.tensor(vw, dtype=torch.float32) / wd,
                                 torch.tensor(vh, dtype=torch.float32) / h])
                    for (h, wd), (vh, vw) in zip(shapes, valid_sizes[b])]))
            c["prop"] = torch.stack(props).contiguous().to(dev)
            c["valid"] = torch.stack(valids).to(torch.uint8).contiguous().to(dev)       # [B, SN]
            c["tok_valid"] = torch.stack(toks).to(torch.uint8).contiguous().to(dev)     # [B, SN]
            c["vr"] = torch.stack(vrs).to(torch.float32).contiguous().to(dev)           # [B, 4, 2]
        c["shapes_dev"] = torch.tensor(shapes, dtype=torch.int64, device=dev)
        c["starts_dev"] = torch.tensor(start, dtype=torch.int64, device=dev)
        c["ready"] = torch.cuda.Event()
        c["ready"].record(torch.cuda.current_stream(dev))
        self._box[key] = c
        while len(self._box) > self.BOX_SHAPES:
            old, _ = self._box.popitem(last=False)
            self._plans.drop(lambda k: (k[0], k[1], k[4]) == old)
        return c

    def _arena(self, slot):
        a = self._arenas.get(slot)
        if a is None:
            a = self._arenas[slot] = Arena(self.device, on_grow=lambda a, s=slot: self._plans.drop(
                lambda k: k[2] == s))
        return a

    def _layout_for(self, dims, own_tokens=True):
        B, shapes = dims[0], [(dims[1 + 2 * l], dims[2 + 2 * l]) for l in range(4)]

        def layout(E):
            pl = CrossHead2._Plan()
            self._layout(pl, E, B, shapes, own_tokens)
            return pl
        return layout

    def _measure(self, dims):
        return measure_bytes(self._layout_for(dims))

    def reserve(self, *a, **k):
        raise NotImplementedError("CrossHeadBBox sizes its arenas on demand")

    def _plan(self, B, shapes, hw2=None, slot=0, nhwc=False, tokens=None):
        """(`hw2`, `nhwc`: CrossHead2's plan signature, unused here -- PipelinedHead calls
        every head the same way.)  A plan is a set of views of the slot's arena (plans.py) plus
        the shared constants of its (shapes, padding); its encoder rows are the neck's token
        buffer when the features are the neck's in-place views (`tokens`)."""
        if tokens is None:
            tokens = getattr(self, "_tokens", None)
        valid_sizes = getattr(self, "_valid_sizes", None)     # None: no padding in this batch
        key = (B, tuple(shapes), slot, None if tokens is None else tokens.data_ptr(), valid_sizes)
        if key in self._plans:
            return self._plans[key]
        if self.w is None:
            self._pack()
        SN = sum(h * wd for h, wd in shapes)
        if SN < self.num_proposals or SN > 65536:
            raise RuntimeError("%d encoder tokens: need between the %d proposals and 65536"
                               % (SN, self.num_proposals))
        dims = (B,) + tuple(v for hw in shapes for v in hw)
        pl = self._arena(slot).carve(self._layout_for(dims, tokens is None), dims, self._measure)
        pl.graph_a = pl.graph_b = pl.graph_cfg = None
        pl.calls_a = pl.calls_b = 0
        pl.streams = {}
        pl.feats_read = torch.cuda.Event()
        if tokens is not None:
            pl.X = tokens
        c = self._box_constants(B, shapes, valid_sizes)
        pl.padded, pl.enc_pos, pl.prop, pl.valid = c["padded"], c["enc_pos"], c["prop"], c["valid"]
        pl.tok_valid, pl.vr = c["tok_valid"], c["vr"]
        pl.shapes_dev, pl.starts_dev, pl.consts_ready = c["shapes_dev"], c["starts_dev"], c["ready"]
        self._plans[key] = pl
        return pl

    def _layout(self, pl, E, B, shapes, own_tokens):
        """Every per-image buffer as a view of the slot's arena (sizes non-decreasing in B and
        in every level's height / width)."""
        i64 = E.i64
        pl.B, pl.shapes = B, list(shapes)
        pl.N = [h * wd for h, wd in shapes]
        pl.start = [sum(pl.N[:l]) for l in range(4)]
        pl.SN = SN = sum(pl.N)
        M, P, K = B * SN, self.num_proposals, self.KEPT
        nc = self.cls_out_channels
        pl.tloc, pl.taw = E(B, SN, 8, 4, 4, 2), E(B, SN, 8, 4, 4)     # (padded batches)
        # ---- encoder ----
        pl.X = E(B, SN, 256)          # (replaced by the neck's token rows when those come in place)
        pl.own_tokens = own_tokens
        pl.X1, pl.Y, pl.S = E(B, SN, 256), E(B, SN, 256), E(B, SN, 256)
        pl.VOA = E(B, SN, 640)
        pl.H = E(M, self.enc_ffn)
        # ---- two-stage ----
        pl.OM = E(M, 256)
        pl.enc_cls = E(B, SN, nc)
        pl.enc_coord, pl.enc_box = E(B, SN, 4), E(B, SN, 4)
        pl.top_idx, pl.top_q, pl.top_r = i64(B, P), i64(B, P), i64(B, P)
        pl.unact = E(B * P, 4)
        pl.ref = [E(B * P, 4) for _ in range(self.num_dec_layers + 1)]
        pl.emb, pl.pt, pl.ptn = E(B * P, 512), E(B * P, 512), E(B * P, 512)
        # ---- decoder ----
        pl.V = [E(B, SN, 256) for _ in range(self.num_dec_layers)]
        BP = B * P
        pl.x, pl.x1, pl.x2, pl.y = E(BP, 256), E(BP, 256), E(BP, 256), E(BP, 256)
        pl.VQKd, pl.attd, pl.OA = E(BP, 768), E(BP, 256), E(BP, 384)
        pl.loc, pl.aw, pl.Sd = E(BP, 8, 4, 4, 2), E(BP, 8, 4, 4), E(BP, 256)
        pl.hd = E(hip.ffn_scratch_floats(BP, self.dec_ffn))
        pl.d1, pl.d2, pl.delta = E(BP, 256), E(BP, 256), E(BP, 4)
        pl.classes = E(B, P, nc)
        pl.qscore = E(B, P)
        pl.keep, pl.keep_q, pl.keep_r = i64(B, K), i64(B, K), i64(B, K)
        pl.cls = E(B, K, nc)
        pl.box = E(B, K, 4)
        # ---- PPN / relation decoder (CrossHead2's buffers) ----
        pl.HW2 = 1
        pl.MP = E(B, K, 1)
        n_keep = pl.N
        pl.N = []
        self._plan_relation(pl, E)
        pl.N = n_keep
        pl.scr = E(max(pl.scr.numel(), hip.attn_scratch_floats(B, P, P)))
        pl.q = E(B * K, 256)
        R = self.num_rel_query
        pl.sub_cls, pl.obj_cls = E(B, R, nc), E(B, R, nc)
        pl.sub_box, pl.obj_box = E(B, R, 4), E(B, R, 4)

    # ------------------------------------------------------------------ stages
    def _encoder(self, pl):
        """6 x [MSDeformAttn over the 4 levels, norm, FFN, norm] on the token rows pl.X."""
        w, B, SN = self.w, pl.B, pl.SN
        X2, X12, Y2 = pl.X.view(-1, 256), pl.X1.view(-1, 256), pl.Y.view(-1, 256)
        for i in range(self.num_enc_layers):
            p = "transformer.encoder.layers.%d." % i
            a = p + "attentions.0."
            hip.gemm(X2, w[a + "voa.weight"], pl.VOA, M=B * SN, N=640, K=256, lda=256, ldw=256,
                     ldc=640, bias=w[a + "voa.bias"], aadd=pl.enc_pos, ldaadd=256,
                     aadd_rows=B * SN if pl.padded else SN, aadd_from_col=256)
            if pl.padded:
                # value rows of padded tokens are zero; the reference points carry the
                # per-image valid ratios: explicit sampling operands + the general sampler
                hip.zero_rows(pl.VOA, pl.tok_valid, pl.VOA, B, SN, 256, ld=640, per_image=True)
                hip.token_sampling(pl.VOA.view(-1)[256:], 640, pl.vr, pl.tloc, pl.taw, B, pl.shapes)
                hip.msda_loc(pl.VOA, 640, pl.shapes_dev, pl.starts_dev, pl.tloc, pl.taw, pl.S, B,
                             SN, SN, 4)
            else:
                hip.msda(pl.VOA, 640, pl.VOA.view(-1)[256:], 640, pl.S, B, pl.shapes)
            # (Linear + identity + LayerNorm as one row-owning launch where `enc_fused_ln` says
            # so, like the pixel decoder's encoder in head.py: bitwise the pair it replaces)
            if "proj" in self.enc_fused_ln:
                hip.linear_res_ln(pl.S.view(-1, 256), w[a + "output_proj.weight"],
                                  w[a + "output_proj.bias"], X2, w[p + "norms.0.weight"],
                                  w[p + "norms.0.bias"], X12)
            else:
                hip.linear(pl.S.view(-1, 256), w[a + "output_proj.weight"],
                           w[a + "output_proj.bias"], Y2, res=X2)
                hip.layernorm(Y2, w[p + "norms.0.weight"], w[p + "norms.0.bias"], X12)
            hip.linear(X12, w[p + "ffns.0.layers.0.0.weight"], w[p + "ffns.0.layers.0.0.bias"],
                       pl.H, relu=True)
            if "ffn" in self.enc_fused_ln:
                hip.linear_res_ln(pl.H, w[p + "ffns.0.layers.1.weight"],
                                  w[p + "ffns.0.layers.1.bias"], X12, w[p + "norms.1.weight"],
                                  w[p + "norms.1.bias"], X2)
            else:
                hip.linear(pl.H, w[p + "ffns.0.layers.1.weight"], w[p + "ffns.0.layers.1.bias"],
                           Y2, res=X12)
                hip.layernorm(Y2, w[p + "norms.1.weight"], w[p + "norms.1.bias"], X2)

    def _box_branch(self, i, src, pl, dst, res=None):
        """reg_branches[i]: Linear-ReLU-Linear-ReLU-Linear(4) (+ res)."""
        w = self.w
        M = src.shape[0]
        d1 = pl.d1 if M == pl.d1.shape[0] else pl.Y.view(-1, 256)
        d2 = pl.d2 if M == pl.d2.shape[0] else pl.S.view(-1, 256)
        hip.linear(src, w["reg_branches.%d.0.weight" % i], w["reg_branches.%d.0.bias" % i], d1,
                   relu=True)
        hip.linear(d1, w["reg_branches.%d.2.weight" % i], w["reg_branches.%d.2.bias" % i], d2,
                   relu=True)
        hip.linear(d2, w["reg_branches.%d.4.weight" % i], w["reg_branches.%d.4.bias" % i], dst,
                   res=res)

    def _two_stage(self, pl):
        """Per-token proposals, their class / box heads, the best `num_proposals` of them and
        the decoder's initial queries; also the decoder layers' value projections of the
        memory (query-independent)."""
        w, B, SN, P = self.w, pl.B, pl.SN, self.num_proposals
        nl, nc = self.num_dec_layers, self.cls_out_channels
        t = "transformer."
        X2 = pl.X.view(-1, 256)
        hip.gemm_group([dict(
            A=X2, W=w[t + "decoder.layers.%d.attentions.1.value_proj.weight" % i], C=pl.V[i],
            bias=w[t + "decoder.layers.%d.attentions.1.value_proj.bias" % i], M=B * SN, N=256,
            K=256, lda=256, ldw=256, ldc=256) for i in range(nl)])
        if pl.padded:    # `value.masked_fill(key_padding_mask)` of the decoder's cross-attentions
            for i in range(nl):
                hip.zero_rows(pl.V[i], pl.tok_valid, pl.V[i], B, SN, 256, per_image=True)
        hip.zero_rows(pl.X, pl.valid, pl.X1, B, SN, 256, per_image=pl.padded)
        hip.linear(pl.X1.view(-1, 256), w[t + "enc_output.weight"], w[t + "enc_output.bias"],
                   pl.Y.view(-1, 256))
        hip.layernorm(pl.Y.view(-1, 256), w[t + "enc_output_norm.weight"],
                      w[t + "enc_output_norm.bias"], pl.OM)
        hip.linear(pl.OM, w["cls_branches.%d.weight" % nl], w["cls_branches.%d.bias" % nl],
                   pl.enc_cls.view(-1, nc))
        self._box_branch(nl, pl.OM, pl, pl.enc_coord.view(-1, 4), res=pl.prop.view(-1, 4))
        hip.sigmoid(pl.enc_coord, pl.enc_box)
        hip.topk_strided(pl.enc_cls, nc, SN * nc, pl.top_idx, pl.top_q, pl.top_r, B, SN, 1, P)
        hip.gather_rows(pl.enc_coord, pl.top_idx, pl.unact, B, SN, P, 4)
        hip.box_pos_embed(pl.unact, pl.ref[0], pl.emb, B * P)
        hip.linear(pl.emb, w[t + "pos_trans.weight"], w[t + "pos_trans.bias"], pl.pt)
        hip.layernorm_rows(pl.pt, w[t + "pos_trans_norm.weight"], w[t + "pos_trans_norm.bias"],
                           pl.ptn)

    def _decoder(self, pl):
        """6 x [self-attention, norm, deformable cross-attention on the reference boxes, norm,
        FFN, norm, box refinement] -> pl.x (last layer's queries), pl.ref[-1], pl.classes."""
        w, B, SN, P = self.w, pl.B, pl.SN, self.num_proposals
        scale = 1.0 / math.sqrt(32.0)
        qpos, x_in = pl.ptn[:, :256], pl.ptn[:, 256:]
        for i in range(self.num_dec_layers):
            p = "transformer.decoder.layers.%d." % i
            sa, ca = p + "attentions.0.attn.", p + "attentions.1."
            hip.linear(x_in, w[sa + "vqk.weight"], w[sa + "vqk.bias"], pl.VQKd, aadd=qpos,
                       aadd_from_col=256)
            hip.attention(pl.VQKd[:, 256:], 768, pl.VQKd[:, 512:], 768, pl.VQKd, 768, None, None,
                          pl.attd, 256, pl.scr, B, P, P, scale)
            hip.linear(pl.attd, w[sa + "out_proj.weight"], w[sa + "out_proj.bias"], pl.y, res=x_in)
            hip.layernorm(pl.y, w[p + "norms.0.weight"], w[p + "norms.0.bias"], pl.x1)
            hip.linear(pl.x1, w[ca + "oa.weight"], w[ca + "oa.bias"], pl.OA, aadd=qpos)
            hip.box_sampling(pl.OA, 384, pl.ref[i], pl.loc, pl.aw, B * P, 4, valid_ratios=pl.vr,
                             rows_per_image=P)
            hip.msda_loc(pl.V[i], 256, pl.shapes_dev, pl.starts_dev, pl.loc, pl.aw, pl.Sd, B, SN,
                         P, 4)
            hip.linear(pl.Sd, w[ca + "output_proj.weight"], w[ca + "output_proj.bias"], pl.y,
                       res=pl.x1)
            hip.layernorm(pl.y, w[p + "norms.1.weight"], w[p + "norms.1.bias"], pl.x2)
            hip.ffn_ln(pl.x2, w[p + "ffns.0.layers.0.0.weight"], w[p + "ffns.0.layers.0.0.bias"],
                       w[p + "ffns.0.layers.1.weight"], w[p + "ffns.0.layers.1.bias"],
                       w[p + "norms.2.weight"], w[p + "norms.2.bias"], pl.x, pl.hd, B * P,
                       self.dec_ffn)
            self._box_branch(i, pl.x, pl, pl.delta)
            hip.box_refine(pl.delta, pl.ref[i], pl.ref[i + 1], B * P)
            x_in = pl.x
        nl = self.num_dec_layers
        hip.linear(pl.x, w["cls_branches.%d.weight" % (nl - 1)], w["cls_branches.%d.bias" % (nl - 1)],
                   pl.classes.view(B * P, -1))

    def _select(self, pl):
        """pairnet_bbox_head.py:252-266: rank the queries (softmax over the query axis, max over
        classes), keep the best 100 in rank order; gather their logits, boxes and features."""
        B, P, K, nc = pl.B, self.num_proposals, self.KEPT, self.cls_out_channels
        hip.query_score(pl.classes, pl.qscore, B, P, nc)
        hip.topk(pl.qscore, pl.keep, pl.keep_q, pl.keep_r, B, P, 1, K)
        hip.gather_rows(pl.classes, pl.keep, pl.cls, B, P, K, nc)
        hip.gather_rows(pl.ref[-1], pl.keep, pl.box, B, P, K, 4)
        hip.gather_rows(pl.x, pl.keep, pl.q, B, P, K, 256)

    def _gather_outputs(self, pl):
        """pairnet_bbox_head.py:322-341."""
        B, K, R, nc = pl.B, self.KEPT, self.num_rel_query, self.cls_out_channels
        hip.gather_rows(pl.cls, pl.sub_pos, pl.sub_cls, B, K, R, nc)
        hip.gather_rows(pl.cls, pl.obj_pos, pl.obj_cls, B, K, R, nc)
        hip.gather_rows(pl.box, pl.sub_pos, pl.sub_box, B, K, R, 4)
        hip.gather_rows(pl.box, pl.obj_pos, pl.obj_box, B, K, R, 4)

    # Stage A: what does not depend on the decoder's query chain -- encoder, per-token heads,
    # proposal selection, the initial queries, the decoder layers' value projections: a few
    # dozen chip-filling launches.  Stage B: the sequential chain (6 decoder layers, query
    # ranking, PPN, top-k, 6 relation layers, gathers).  Same split as CrossHead2, so
    # `PipelinedHead` schedules this head unchanged.
    def _stage_a(self, feats, pl):
        if pl.own_tokens:      # any layout but the neck's in-place views: copy into token rows
            self._stage_a_copy(feats, pl)
        self._encoder(pl)
        self._two_stage(pl)

    def _stage_b(self, pl):
        self._decoder(pl)
        self._select(pl)
        self._pair_proposal(pl)
        self._relation_decoder(pl)

    def _run_stage(self, which, pl, feats=None):
        """Stage 'a' or 'b' of plan `pl` on the current stream: eagerly, or (with `use_graphs`,
        from the second call on, at quiet points only: plans.quiet) as one hipGraph replay.  A
        stage-A graph is tied to the buffers it was captured on: the neck's in-place token rows belong to the plan (its
        key), plain feature tensors are staged through the plan's own token rows."""
        cfg = (self.fuse_ppn_front, self.grid_reserve, tuple(self.enc_fused_ln))
        if pl.graph_cfg != cfg:
            pl.graph_a = pl.graph_b = None
            pl.graph_cfg = cfg
        cur = torch.cuda.current_stream(self.device)
        pl.streams[cur.cuda_stream] = cur
        try:
            self._run_stage_on(which, pl, feats, cur)
        finally:
            plans.note_use(cur)

    def _run_stage_on(self, which, pl, feats, cur):
        if which == "a":
            if not pl.consts_ready.query():     # (shared constants filled on another stream)
                cur.wait_event(pl.consts_ready)
            if pl.own_tokens:
                self._stage_a_copy(feats, pl)
                pl.feats_read.record()       # the caller's feature buffers are free again
            def body():
                with hip.reserve_slots(self.grid_reserve):
                    self._encoder(pl)
                    self._two_stage(pl)
            if self.use_graphs and pl.graph_a is None and pl.calls_a >= 1 and plans.quiet(cur):
                pl.graph_a = self._capture(body)
            pl.calls_a += 1
            if self.use_graphs and pl.graph_a is not None:
                pl.graph_a.replay()
            else:
                body()
            if not pl.own_tokens:
                pl.feats_read.record()       # (the neck's token rows are encoded in place)
        else:
            if self.use_graphs and pl.graph_b is None and pl.calls_b >= 1 and plans.quiet(cur):
                pl.graph_b = self._capture(lambda: self._stage_b(pl))
            pl.calls_b += 1
            if self.use_graphs and pl.graph_b is not None:
                pl.graph_b.replay()
            else:
                self._stage_b(pl)

    def _stage_a_copy(self, feats, pl):
        B = pl.B
        for f, s, n, (h, w) in zip(feats, pl.start, pl.N, pl.shapes):
            pl.X[:, s:s + n].view(B, h, w, 256).permute(0, 3, 1, 2).copy_(f)

    def _outputs(self, pl):
        return (dict(sub=pl.sub_cls, obj=pl.obj_cls, cls=pl.cls, enc_cls_scores=pl.enc_cls,
                     enc_bbox_preds=pl.enc_box, rel=pl.rel, importance=pl.imp),
                dict(bbox=pl.box, sub_bbox=pl.sub_box, obj_bbox=pl.obj_box))

    # ----------------------------------------------------------------- forward
    def _check_feats(self, feats, img_metas):
        B = len(img_metas)
        if len(feats) != 4 or any(f.shape[0] != B or f.shape[1] != 256 for f in feats):
            raise RuntimeError("mlvl_feats: the neck's 4 levels of [B, 256, h, w]")
        for f in feats:
            if not f.is_cuda or f.dtype != torch.float32:
                raise RuntimeError("mlvl_feats must be fp32 device tensors")
        ih, iw = img_metas[0].get("batch_input_shape", img_metas[0]["img_shape"][:2])
        shapes = [tuple(f.shape[-2:]) for f in feats]
        self._valid_sizes = None
        if any(tuple(m["img_shape"][:2]) != (ih, iw) for m in img_metas):
            # a padded batch (pairnet_bbox_head.py:196-213): per-image key-padding masks
            metas = [dict(m, batch_input_shape=(ih, iw)) for m in img_metas]
            self._valid_sizes = self.level_valid_sizes(metas, shapes)
        if self.device is None:
            self.to(feats[0].device)
        self._feats_nhwc = False
        self._tokens = self._token_buffer(feats, shapes)
        return B, shapes, None

    @staticmethod
    def _token_buffer(feats, shapes):
        """The neck's outputs are views of one [B, SN, 256] token buffer (neck.py): find it."""
        B = feats[0].shape[0]
        SN = sum(h * w for h, w in shapes)
        base, off = feats[0].data_ptr(), 0
        for f, (h, w) in zip(feats, shapes):
            if f.data_ptr() != base + off * 256 * 4 or \
                    tuple(f.stride()) != (SN * 256, 1, w * 256, 256):
                return None
            off += h * w
        return torch.as_strided(feats[0], (B, SN, 256), (SN * 256, 256, 1))

    @torch.no_grad()
    @hip.on_device
    def forward(self, mlvl_feats, img_metas, slot=0):
        """mlvl_feats: the neck's four (B, 256, h, w) levels, high -> low resolution, fp32 on
        the GPU; returns the reference's two dicts (pairnet_bbox_head.py:343-359).  Output
        tensors are views of per-shape buffers that the next forward() of the same shape (and
        slot) overwrites."""
        B, shapes, _ = self._check_feats(mlvl_feats, img_metas)
        pl = self._plan(B, shapes, None, slot)
        self._run_stage("a", pl, mlvl_feats)
        self._run_stage("b", pl)
        self._last_plan = pl
        return self._outputs(pl)

    __call__ = forward

    def forward_head(self, *a, **k):
        raise NotImplementedError("CrossHeadBBox has no forward_head (pairnet_bbox_head.py)")

    # ------------------------------------------------------- post-processing
    @torch.no_grad()
    @hip.on_device
    def get_bboxes(self, cls_scores, bbox_preds, img_metas, rescale=False):
        """pairnet_bbox_head.py:1012-1041 (the two unused arguments the reference slices with
        `[-1, img_id]`, :1016-1017, are not read)."""
        self._pan_jobs = []
        return [self._get_bboxes_single(
            cls_scores["sub"][i], cls_scores["obj"][i], cls_scores["rel"][i],
            bbox_preds["sub_bbox"][i], bbox_preds["obj_bbox"][i], img_metas[i]["img_shape"],
            img_metas[i]["scale_factor"], rescale) for i in range(len(img_metas))]

    def _get_bboxes_single(self, s_cls, o_cls, r_cls, s_box, o_box, img_shape, scale_factor,
                           rescale=False):
        """pairnet_bbox_head.py:1043-1101 on the device, asynchronously: returns the
        reference's 6-tuple (det_bboxes, labels, rel_pairs, r_scores, r_labels, r_dists)."""
        assert len(s_cls) == len(o_cls) == len(r_cls)
        dev, R, nc = s_cls.device, s_cls.shape[0], s_cls.shape[-1]
        pb = self._post_buffers(s_cls, ("bbox", R), lambda: dict(
            det=torch.empty(2 * R, 5, device=dev), labels=torch.empty(2 * R, device=dev,
                                                                       dtype=torch.int64),
            r_dists=torch.empty(R, self.num_relations + 1, device=dev),
            pairs=torch.arange(2 * R, dtype=torch.int).reshape(2, -1).T,
            zeros=torch.zeros(100, device=dev)))
        sf = list(scale_factor) if hasattr(scale_factor, "__len__") else [scale_factor] * 4
        hip.box_triplets(s_cls.contiguous(), o_cls.contiguous(), s_box.contiguous(),
                         o_box.contiguous(), pb["det"], pb["labels"], R, nc, img_shape[0],
                         img_shape[1], sf, rescale)
        hip.rel_dists(r_cls.contiguous(), pb["r_dists"], R, self.num_relations)
        return (pb["det"], pb["labels"], pb["pairs"], pb["zeros"], pb["zeros"], pb["r_dists"])

    def panoptic_status(self, results=None):
        return dict(active=0, all_gone=0, rounds=0)
