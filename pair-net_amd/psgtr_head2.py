"""PSGTrHead2 on MI355X: the reference's "PSGTR on Mask2Former" sibling head (every query
is one triplet with subject / object / predicate heads) on the shared trunk of `CrossHead2`.

Mirrors pairnet/models/relation_heads/psgtr_head2.py: constructor keywords (:24-86),
`forward(feats, img_metas)` (:345-444), `get_bboxes` (:1063-1152), `simple_test_bboxes`
(:1154-1158) and the 308 state-dict key names, so `configs/psgtr/psgtr_r50_psg_plus.py`
and its checkpoint drop in.  Same kernels, same C ABI, no CPU path.

Reference behaviour kept on purpose:
  * the loop at :404-411 unpacks forward_head's outputs as `..., mask_pred_sub,
    mask_pred_sub, attn_mask`: the name `mask_pred_sub` ends up holding the FIFTH output,
    so the returned `sub_seg` is obj_mask_embed(post_norm(q_last)) . mask_feature, the
    returned `obj_seg` is the object mask of the INITIAL forward_head call,
    obj_mask_embed(post_norm(query_feat)) . mask_feature, and `sub_mask_embed` never
    reaches an output (its weights are still part of the checkpoint layout);
  * outputs carry the reference's leading stacked-layer dimension of size 1
    (`torch.stack(...)[-1:]`, :433-441).
Extension: the reference's get_bboxes indexes that size-1 dimension with the image id, so
it only works for one image per call; here batches of any size are post-processed per
image (identical for batch 1).
"""
import torch

from . import hip
from .head import CrossHead2


class PSGTrHead2(CrossHead2):
    """Drop-in for the reference's `PSGTrHead2` (inference half)."""

    def __init__(self, num_classes, num_relations, in_channels=(256, 512, 1024, 2048),
                 use_mask=True, num_obj_query=100, num_reg_fcs=2, n_heads=8, embed_dims=256,
                 swin_backbone=None, sync_cls_avg_factor=False, bg_cls_weight=0.02,
                 sub_loss_cls=None, sub_loss_mask=None, sub_loss_dice=None, obj_loss_cls=None,
                 obj_loss_mask=None, obj_loss_dice=None, rel_loss_cls=None, train_cfg=None,
                 **kwargs):
        kwargs.pop("num_rel_query", None)
        kwargs.setdefault("relation_decoder", dict(
            num_layers=0, transformerlayers=dict(ffn_cfgs=dict(feedforward_channels=2048))))
        super().__init__(num_classes, list(in_channels), num_relations,
                         num_obj_query=num_obj_query, num_rel_query=num_obj_query,
                         use_mask=use_mask, n_heads=n_heads, embed_dims=embed_dims,
                         train_cfg=None, **kwargs)

    # ------------------------------------------------------------------ params
    def param_shapes(self):
        """psgtr_head2.py:204-252 in registration order."""
        s = super().param_shapes()
        for k in list(s):
            if k.startswith(("relation_decoder.", "rel_query_", "update_importance.", "cls_embed.",
                             "sub_query_update.", "obj_query_update.", "rel_cls_embed.",
                             "mask_embed.")):
                del s[k]
        nc = self.num_classes + 1
        for name, n in (("sub_cls_embed", nc), ("obj_cls_embed", nc),
                        ("rel_cls_embed", self.num_relations + 1)):
            s[name + ".weight"] = (n, 256)
            s[name + ".bias"] = (n,)
        for mlp in ("sub_mask_embed", "obj_mask_embed", "mask_embed"):
            for j in (0, 2, 4):
                s["%s.%d.weight" % (mlp, j)] = (256, 256)
                s["%s.%d.bias" % (mlp, j)] = (256,)
        return s

    def _pack_relation(self, w):
        for i in range(self.num_dec_layers):
            self._pack_vqk(w, "transformer_decoder.layers.%d.attentions.1.attn." % i)

    def _plan_relation(self, pl, E):
        B, Q = pl.B, self.num_obj_query
        nc = self.num_classes + 1
        pl.scr = E(max(hip.attn_scratch_floats(B, Q, n) for n in pl.N + [Q]))
        pl.sub_cls, pl.obj_cls = E(1, B, Q, nc), E(1, B, Q, nc)
        pl.rel = E(1, B, Q, self.num_relations + 1)
        pl.sub_seg = pl.MP.view(1, B, Q, pl.HW2)     # last layer's masks
        pl.obj_seg = E(1, B, Q, pl.HW2)
        pl.me2 = E(B * Q, 256)

    # ------------------------------------------------------------------ stages
    def _stage_b(self, pl):
        w, B, Q = self.w, pl.B, self.num_obj_query
        # the stale object mask (:404-411): obj_mask_embed on the post-normed initial queries
        hip.layernorm(pl.q0, w["transformer_decoder.post_norm.weight"],
                      w["transformer_decoder.post_norm.bias"], pl.qn)
        self._mlp3("obj_mask_embed", pl.qn, pl.me2, pl)
        self._mask_logits(pl.me2, pl, pl.obj_seg.view(B, Q, pl.HW2))
        self._object_decoder(pl, final_head=False)
        # heads of the last layer (:309-324)
        hip.layernorm(pl.q, w["transformer_decoder.post_norm.weight"],
                      w["transformer_decoder.post_norm.bias"], pl.qn)
        for name, dst in (("sub_cls_embed", pl.sub_cls), ("obj_cls_embed", pl.obj_cls),
                          ("rel_cls_embed", pl.rel)):
            hip.linear(pl.qn, w[name + ".weight"], w[name + ".bias"], dst.view(B * Q, -1))
        self._mlp3("obj_mask_embed", pl.qn, pl.me2, pl)      # sic: see the module docstring
        self._mask_logits(pl.me2, pl, pl.MP)

    def _outputs(self, pl):
        B, Q = pl.B, self.num_obj_query
        H2, W2 = pl.hw2
        return (dict(sub=pl.sub_cls, obj=pl.obj_cls, rel=pl.rel),
                dict(sub_seg=pl.sub_seg.view(1, B, Q, H2, W2),
                     obj_seg=pl.obj_seg.view(1, B, Q, H2, W2)))

    def pair_positions(self, pl=None):
        """Query i IS triplet i (psgtr_head2.py:345-444): identity rows."""
        pl = pl if pl is not None else self._last_plan
        ident = self.__dict__.get("_ident")
        if ident is None or ident.device != self.device:
            # a constant, made ONCE and complete before it is handed out: consumers read it on
            # other streams (the chain stream of a pipelined result) without any ordering
            # against the stream it was created on
            ident = torch.arange(self.num_obj_query, device=self.device, dtype=torch.int64)
            torch.cuda.current_stream(self.device).synchronize()
            self._ident = ident
        return ident.unsqueeze(0).expand(pl.B, -1), ident.unsqueeze(0).expand(pl.B, -1)

    def forward_head(self, decoder_out, mask_feature, attn_mask_target_size):
        raise NotImplementedError("PSGTrHead2.forward_head (six outputs, psgtr_head2.py:288) is "
                                  "internal to forward() here")

    # ------------------------------------------------------- post-processing
    @torch.no_grad()
    @hip.on_device
    def get_bboxes(self, cls_scores, mask_preds, img_metas, rescale=False):
        """psgtr_head2.py:1063-1085 (per image; see the module docstring)."""
        self._pan_jobs = []          # (pan_img is the constant map of :1129: no device loop)
        return [self._get_bboxes_single(
            cls_scores["sub"][0, i], cls_scores["obj"][0, i], cls_scores["rel"][0, i],
            mask_preds["sub_seg"][0, i], mask_preds["obj_seg"][0, i],
            img_metas[i]["img_shape"], img_metas[i]["scale_factor"], rescale)
            for i in range(len(img_metas))]

    def _get_bboxes_single(self, s_cls, o_cls, r_cls, s_seg, o_seg, img_shape, scale_factor,
                           rescale=False):
        """psgtr_head2.py:1087-1152 on the device, asynchronously."""
        assert len(s_cls) == len(o_cls) == len(r_cls)
        dev = s_cls.device
        Q, nc, nrel1 = s_cls.shape[0], s_cls.shape[-1], r_cls.shape[-1]
        H0 = round(img_shape[0] / scale_factor[1])
        W0 = round(img_shape[1] / scale_factor[0])
        h, wd = s_seg.shape[-2:]
        labels = torch.empty(2 * Q, device=dev, dtype=torch.int64)
        sc_tmp = torch.empty(2 * Q, device=dev, dtype=torch.float32)
        hip.cls_argmax(s_cls.contiguous(), labels[:Q], sc_tmp[:Q], Q, nc, 1)
        hip.cls_argmax(o_cls.contiguous(), labels[Q:], sc_tmp[Q:], Q, nc, 1)
        r_dists = torch.empty(Q, nrel1, device=dev, dtype=torch.float32)
        fg = torch.empty(Q * (nrel1 - 1), device=dev, dtype=torch.float32)
        hip.softmax_fg(r_cls.contiguous(), r_dists, fg, Q, nrel1)
        masks_u8 = torch.empty(2 * Q, H0, W0, device=dev, dtype=torch.uint8)
        hip.bilinear_planar_gt0(s_seg.contiguous(), masks_u8[:Q], Q, h, wd, H0, W0)
        hip.bilinear_planar_gt0(o_seg.contiguous(), masks_u8[Q:], Q, h, wd, H0, W0)
        pan_img = torch.ones((H0, W0), device=dev, dtype=torch.int64)     # :1129
        return (torch.zeros((2 * Q, 5), device=dev), labels,
                torch.arange(2 * Q, dtype=torch.int).reshape(2, -1).T, masks_u8.view(torch.bool),
                pan_img, torch.zeros(Q, device=dev), torch.zeros(Q, device=dev), r_dists)
