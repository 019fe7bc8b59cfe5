"""One training iteration of Pair-Net's own parameters on the MI355X: the
reference's `forward_train` -> `loss` -> `backward` -> clip -> AdamW step (frameworks/psgtr.py:112-146,
pairnet_head.py:419-757, tools/train.py:115-241, configs/mask2former/pairnet.py:353-368) for the
parameters of `CrossHead2` the loss reaches: by default the Relation Fusion decoder,
`rel_cls_embed`, the relation query / position embeddings, the Pair Proposal Network's MLPs and the
Matrix Learner (10.2 M), with `train_decoder` the nine masked decoder layers as well (+14.3 M) and
with `train_pixel_decoder` the pixel decoder's encoder path (+5.3 M), with `backbone=` the ResNet's
stages 2-4 (+23.2 M; stem, layer1 and BatchNorm frozen as in the reference's config) -- 53.0 M, the
reference's whole trainable graph.  Without `backbone=` the backbone is frozen (a fine-tuning
regime); a Swin backbone has no backward here.
Everything the reference's loss can reach enters through two logits: `loss_r_cls` through
`rel`, `loss_match` through `importance` (`loss_sub_cls` / `loss_obj_cls` read DETACHED class
logits, pairnet_head.py:380-390, and train nothing).

    trainer = TailTrainer(head)                       # AdamW(lr=1e-4, wd=1e-4), clip 0.1
    losses = trainer.step(feats, img_metas, gt_rels, gt_labels, gt_masks)

Per step: `head.forward` (the inference kernels; hipGraphs if on) -> `head.loss(grads=)` (csrc/
loss.hip; the two Hungarian assignments on the host as the reference) -> `RelationTailGrad`
(/ `HeadGrad` / `PixelDecoderGrad` / `BackboneGrad`) forward-with-tape + backward into one flat
gradient buffer -> (world > 1) bucketed all-reduce
overlapped with the backward pass (`dist.GradReducer`) -> global-norm clip coefficient on the
device -> ONE AdamW launch over the flat parameter buffer that the head's weight dict aliases ->
refresh of the derived weight packs the inference kernels read.  No host synchronisation inside a
step apart from the reference's own (the Hungarian cost matrices).
"""
from collections import OrderedDict

import torch

from . import hip
from .dist import GradReducer
from .grad import BackboneGrad, HeadGrad, PixelDecoderGrad, RelationTailGrad

__all__ = ["TailTrainer", "step_lr"]


def step_lr(base_lr, epoch, steps=(5, 10), gamma=0.5):
    """mmcv's StepLrUpdaterHook by epoch (`lr_config = dict(policy="step", gamma=0.5, step=[5, 10])`,
    configs/mask2former/pairnet.py:370): the learning rate of 0-based `epoch`."""
    return base_lr * gamma ** sum(1 for s in steps if epoch >= s)


class TailTrainer:
    FROZEN_GROUPS = ("cls",)      # cls_embed / post_norm: no gradient in the reference's graph

    def __init__(self, head, lr=1e-4, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8,
                 max_norm=0.1, norm_decay_mult=0.0, lr_mult=None, group=None,
                 bucket_bytes=32 << 20, train_decoder=False, train_pixel_decoder=False,
                 backbone=None):
        """`train_decoder`: also train the nine masked decoder layers, `query_feat`, `query_embed`
        and `level_embed` (`HeadGrad`; the reference's `transformer_decoder` group, lr_mult 0.1 by
        default here as in configs/mask2former/pairnet.py:358-363) -- everything of the head behind
        the pixel decoder.  `lr_mult`: {substring of a parameter name: multiplier} (mmcv's
        `paramwise_cfg.custom_keys`)."""
        self.head = head
        # `backbone` (a ResNet50Hip): also train its stages 2-4 (`BackboneGrad`; the reference's
        # `backbone` group at lr_mult 0.1, stem / layer1 / BatchNorm frozen as in its config):
        # `step()` then takes the IMAGE tensor.  Its parameters appear as "backbone.<name>".
        self.backbone = backbone
        train_pixel_decoder = bool(train_pixel_decoder) or backbone is not None
        self.train_pixel_decoder = bool(train_pixel_decoder)
        self.train_decoder = bool(train_decoder) or self.train_pixel_decoder
        train_decoder = self.train_decoder
        if lr_mult is None:
            lr_mult = {"transformer_decoder": 0.1, "pixel_decoder": 0.1, "backbone.": 0.1}
        if head.w is None:
            head._pack()
        # plans captured so far bake the addresses of the weight tensors that are re-homed below
        # into their hipGraphs: start from fresh plans (the arenas, shape-dependent only, stay)
        from .plans import PlanCache
        head._plans = PlanCache(head._plans.max_plans)
        dev = self.dev = head.device
        tape_cls = HeadGrad if train_decoder else RelationTailGrad
        # ONE flat gradient buffer for every tape: [head tape | pixel decoder tape]
        n_head = tape_cls.size_of(head)
        n_pd = PixelDecoderGrad.size_of(head) if self.train_pixel_decoder else 0
        if backbone is not None:
            if backbone.device is None:
                backbone.to(dev)
            if backbone.w is None:
                backbone._pack()
        n_bb = BackboneGrad.size_of(backbone) if backbone is not None else 0
        self.flat_grad = torch.zeros(n_head + n_pd + n_bb, device=dev, dtype=torch.float32)
        self.tape = tape = tape_cls(head, flat=self.flat_grad, base=0)
        self.pd_tape = PixelDecoderGrad(head, flat=self.flat_grad, base=n_head) \
            if self.train_pixel_decoder else None
        self.bb_tape = BackboneGrad(backbone, flat=self.flat_grad, base=n_head + n_pd) \
            if backbone is not None else None
        self.lr, self.wd, self.betas, self.eps, self.max_norm = lr, weight_decay, betas, eps, max_norm
        # name -> (offset in the shared buffer, shape, numel); the class path (cls_embed /
        # post_norm: no gradient in the reference's graph) stays in the layout with lr 0
        self.layout = OrderedDict(tape.layout)
        if self.pd_tape is not None:
            for n, (o, shape, k) in self.pd_tape.layout.items():
                self.layout[n] = (o + n_head, shape, k)
        src = {n: head._params[n] for n in self.layout}          # where a parameter's value lives
        if self.bb_tape is not None:
            for n, (o, shape, k) in self.bb_tape.layout.items():
                self.layout["backbone." + n] = (o + n_head + n_pd, shape, k)
                src["backbone." + n] = backbone._params[n]
        frozen = {n for g, names in tape.param_groups(head) if g in self.FROZEN_GROUPS for n in names}
        self.n = n_head + n_pd + n_bb
        self.names = [n for n in self.layout if n not in frozen]
        self._frozen = frozen
        # ---- flat parameters; the head's device weights become views of them ----
        self.flat_p = torch.zeros(self.n, device=dev, dtype=torch.float32)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        w = head.w
        self.params = OrderedDict()
        ml = "update_importance.conv_layers."
        self._repacked = {ml + "1.0.weight", ml + "2.0.weight"}   # w[...] holds another layout
        for n in self.layout:
            o, shape, k = self.layout[n]
            view = self.flat_p[o:o + k].view(shape)
            view.copy_(src[n].to(dev))
            if n in frozen:
                continue
            self.params[n] = view
            if n.startswith("backbone."):
                continue          # the backbone's packed weights are all derived (BatchNorm folded)
            if n not in self._repacked:
                # same bytes as the reference layout (ConvTiny's first layer: [64,1,7,7] == [64,49],
                # the 1x1 input convolutions: [256,C,1,1] == [256,C])
                w[n] = view.view(w[n].shape)
        # ---- per-segment multipliers (mmcv paramwise_cfg) ----
        lr_mult = dict(lr_mult or {})
        offs, lrs, wds = [], [], []
        for n in self.layout:
            o, shape, k = self.layout[n]
            offs.append(o)
            lrs.append(0.0 if n in frozen else
                       next((m for key, m in lr_mult.items() if key in n), 1.0))
            is_norm = ".norms." in n or "post_norm" in n or ".gn." in n
            wds.append(0.0 if n in frozen else (norm_decay_mult if is_norm else 1.0))
        offs.append(self.n)
        self.seg_off = torch.tensor(offs, dtype=torch.int64, device=dev)
        self.seg_lr = torch.tensor(lrs, dtype=torch.float32, device=dev)
        self.seg_wd = torch.tensor(wds, dtype=torch.float32, device=dev)
        self.clip = torch.zeros(2, device=dev, dtype=torch.float32)     # [grad norm, coefficient]
        self._scratch = torch.zeros(256, device=dev, dtype=torch.float64)
        self.reducer = GradReducer(self.flat_grad, group=group, bucket_bytes=bucket_bytes)
        self.steps = 0
        self._w_dicts = (head.w, backbone.w if backbone is not None else None)
        self._refresh_derived()

    # ------------------------------------------------------------------
    def _refresh_derived(self):
        """The weight packs the INFERENCE kernels read that are functions of trained parameters,
        rewritten in place (captured hipGraphs keep their pointers)."""
        head, w, p = self.head, self.head.w, self.params
        decs = [("relation_decoder", head.num_rel_layers)]
        if self.train_decoder:
            decs.append(("transformer_decoder", head.num_dec_layers))
        for dec, nl in decs:                       # self-attention as one [V | Q | K] projection
            for i in range(nl):
                a = "%s.layers.%d.attentions.1.attn." % (dec, i)
                W, b = p[a + "in_proj_weight"], p[a + "in_proj_bias"]
                w[a + "vqk.weight"][:256].copy_(W[512:])
                w[a + "vqk.weight"][256:].copy_(W[:512])
                w[a + "vqk.bias"][:256].copy_(b[512:])
                w[a + "vqk.bias"][256:].copy_(b[:512])
        for i in range(head.num_rel_layers):       # relation cross-attention keys / values: [V | K]
            a = "relation_decoder.layers.%d.attentions.0.attn." % i
            W, b = p[a + "in_proj_weight"], p[a + "in_proj_bias"]
            w[a + "vk.weight"][:256].copy_(W[512:])
            w[a + "vk.weight"][256:].copy_(W[256:512])
            w[a + "vk.bias"][:256].copy_(b[512:])
            w[a + "vk.bias"][256:].copy_(b[256:512])
        ml = "update_importance.conv_layers."
        w[ml + "1.0.weight"].copy_(p[ml + "1.0.weight"].permute(0, 2, 3, 1).reshape(64, -1))
        w[ml + "2.0.weight"].copy_(p[ml + "2.0.weight"].reshape(64, 49).t())
        for c in head._consts.values():           # rel_query_feat repeated over the batch
            if "r0" in c:
                B = c["r0"].shape[0] // p["rel_query_feat.weight"].shape[0]
                c["r0"].view(B, -1, 256).copy_(p["rel_query_feat.weight"].unsqueeze(0).expand(B, -1, -1))
        if self.train_decoder:
            from types import SimpleNamespace
            for c in head._consts.values():       # the initial queries and their mask embedding
                if "q0" not in c:
                    continue
                BQ = c["q0"].shape[0]
                B = BQ // p["query_feat.weight"].shape[0]
                c["q0"].view(B, -1, 256).copy_(p["query_feat.weight"].unsqueeze(0).expand(B, -1, -1))
                if c.get("me0") is not None:
                    tmp = SimpleNamespace(B=B, cls=None, MP=None,
                                          **{n: torch.empty(BQ, 256, device=self.dev)
                                             for n in ("qn", "m1", "m2", "me")})
                    head._head_embed(c["q0"], tmp, False, False)
                    c["me0"].copy_(tmp.me)
            for shapes, ent in head._pe.items():  # key position tables carry level_embed
                for l, (h, wd) in enumerate(shapes):
                    hip.sine_pe(ent[1][l], p["level_embed.weight"][l], h, wd)
        if self.train_pixel_decoder:
            pd = "pixel_decoder."
            for i in range(head.num_enc_layers):
                q, a = pd + "encoder.layers.%d." % i, pd + "encoder.layers.%d.attentions.0." % i
                # one [value_proj | sampling_offsets | attention_weights] projection per layer
                o = 0
                for n in ("value_proj", "sampling_offsets", "attention_weights"):
                    r = p[a + n + ".weight"].shape[0]
                    w[a + "voa.weight"][o:o + r].copy_(p[a + n + ".weight"])
                    w[a + "voa.bias"][o:o + r].copy_(p[a + n + ".bias"])
                    o += r
                for k in (a + "voa.weight", a + "output_proj.weight", q + "ffns.0.layers.0.0.weight",
                          q + "ffns.0.layers.1.weight"):
                    hip.s3_split(w[k], w[k + ".s3"])          # the bf16-pipe GEMMs' operands
            for shapes, ent in head._pe.items():  # query position tables carry level_encoding
                o = 0
                for l, (h, wd) in enumerate(shapes):
                    hip.sine_pe(ent[0][o:o + h * wd], p[pd + "level_encoding.weight"][l], h, wd)
                    o += h * wd
                ent[3].copy_(hip.pos8(ent[0]))
        if self.bb_tape is not None:
            self._refresh_backbone()

    def _refresh_backbone(self):
        """Re-fold BatchNorm into the trained convolutions and rewrite the backbone's packed
        weights (channel-last rows, Winograd transforms, bf16-plane splits) in place."""
        bb, p = self.backbone, self.params
        w = bb.w
        for n, sc in self.bb_tape.bn_scale64.items():
            conv = n[:-len(".weight")]
            # W' = W gamma / sqrt(var + eps), formed in double and rounded once like `_fold`
            cw = (p["backbone." + n].double() * sc.view(-1, 1, 1, 1)).float()
            co = cw.shape[0]
            w[conv + ".w"].copy_(cw.permute(0, 2, 3, 1).reshape(co, -1))
            if conv + ".wino" in w:
                w[conv + ".wino"].copy_(hip.winograd_weights(cw.contiguous()))
                w[conv + ".wino4"].copy_(hip.winograd43_weights(cw.contiguous()))
            if conv + ".w.s3" in w:
                hip.s3_split(w[conv + ".w"], w[conv + ".w.s3"])

    def set_epoch(self, epoch, steps=(5, 10), gamma=0.5):
        """The reference's learning-rate schedule (step policy by epoch): sets `self.lr`."""
        if not hasattr(self, "base_lr"):
            self.base_lr = self.lr
        self.lr = step_lr(self.base_lr, epoch, steps, gamma)
        return self.lr

    def write_back(self):
        """Copy the trained values into the head's (and the backbone's) state dict (checkpoints,
        `state_dict()`)."""
        head = self.head
        for n, v in self.params.items():
            dst = self.backbone._params[n[len("backbone."):]] if n.startswith("backbone.") \
                else head._params[n]
            dst.copy_(v.to(dst.device))
        head._packed_version = head._weights_version()     # the packed device weights ARE these values
        if self.backbone is not None:
            self.backbone._packed_version = self.backbone._weights_version()

    # ------------------------------------------------------------------
    @torch.no_grad()
    @hip.on_device
    def step(self, feats, img_metas, gt_rels, gt_labels, gt_masks, point_coords=None):
        """One iteration on one batch; returns the four loss terms (device scalars, as
        `CrossHead2.loss`) plus `grad_norm` (device scalar, before clipping).  `feats`: the four
        backbone feature maps -- or, for a trainer built with `backbone=`, the image tensor."""
        head, tape = self.head, self.tape
        if head.w is not self._w_dicts[0] or (self.backbone is not None
                                              and self.backbone.w is not self._w_dicts[1]):
            # load_state_dict / an in-place parameter update made the module re-pack its device
            # weights: the flat buffers of this trainer no longer back them
            raise RuntimeError("the head (or backbone) re-packed its weights since this TailTrainer "
                               "was built (load_state_dict, parameter update, .to()): build a new one")
        if self.backbone is not None:
            feats = [f.clone(memory_format=torch.preserve_format) for f in self.backbone(feats)]
        outs = head.forward(feats, img_metas)
        up = {}
        losses = head.loss(*outs, gt_rels, None, gt_labels, gt_masks, img_metas,
                           point_coords=point_coords, grads=up)
        pl = head._last_plan
        if self.train_decoder:
            tape.forward_from_plan(pl, pl.sub_pos, pl.obj_pos)
        else:
            tape.forward(pl.q, pl.sub_pos, pl.obj_pos)
        if self.pd_tape is not None:
            self.pd_tape.forward(feats)
        self.reducer.start()
        nh = tape.flat_numel
        back = tape.backward(g_rel=up["rel"], g_importance=up["importance"],
                             on_ready=lambda e: self.reducer.ready(min(e, nh)))
        if self.pd_tape is not None:           # back[0]: d memory tokens (HeadGrad)
            npd = self.pd_tape.flat_numel
            dfeats, _ = self.pd_tape.backward(back[0], on_ready=lambda e: self.reducer.ready(nh + e))
            if self.bb_tape is not None:       # dfeats: d C5, d C4, d C3
                self.bb_tape.forward(feats[0])
                self.bb_tape.backward(dfeats[2], dfeats[1], dfeats[0],
                                      on_ready=lambda e: self.reducer.ready(nh + npd + e))
        self.reducer.finish()
        self.apply_gradients()
        out = dict(losses)
        out["grad_norm"] = self.clip[0]
        return out

    @torch.no_grad()
    @hip.on_device
    def apply_gradients(self):
        """Clip (global L2 norm over the trainable gradients) + AdamW on `tape.flat_grad`."""
        g = self.flat_grad
        pre = self.reducer.scale
        hip.grad_norm_clip(g, self.clip, self._scratch, pre=pre, max_norm=self.max_norm)
        self.steps += 1
        hip.adamw(self.flat_p, g, self.flat_m, self.flat_v, self.seg_off, self.seg_lr, self.seg_wd,
                  self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.steps,
                  clip=self.clip, pre=pre)
        self._refresh_derived()
