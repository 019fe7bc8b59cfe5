"""Backward of Pair-Net's own tail on the MI355X (second slice; no optimizer, no
DDP): given the object decoder's output queries `q` and the gradients of the four loss terms
with respect to the head's logits (`CrossHead2.loss(..., grads={})`, csrc/loss.hip), the
gradients of everything between them --

  * the Relation Fusion decoder x6 + `rel_cls_embed` (pairnet_head.py:353-378; post-norm layers
    in the order cross-attention, norm, self-attention, norm, FFN, norm: facebook_detr.py:378-432),
  * the pair-feature gather (:342-351) back onto the query rows,
  * the Pair Proposal Network: `sub_query_update` / `obj_query_update` MLPs, F.normalize, the
    cosine matrix (:322-333) and the Matrix Learner ConvTiny (frameworks/cnn_factory.py:6-53),
  * optionally the subject / object class gathers (:380-390) through `cls_embed` and `post_norm`
    (:236-238) -- the reference DETACHES `cls_pred` there, so by default they carry no gradient

-- with respect to `q` and to every parameter on the way, named and laid out as the reference's
state dict.  In the reference this is `torch.autograd` behind `losses.backward()`; here the
forward is re-run with the product's forward kernels into a tape of per-layer buffers
(`forward`), and `backward` walks it with csrc/grad.hip's kernels plus `pn_gemm_f32` (a linear
layer's dX = dY W and dW = dY^T X are GEMMs on transposed operands) and `pn_conv2d_nhwc_ex_f32`
(the 64 -> 64 convolution's data gradient).  The top-k pair selection is piecewise constant: the
selected (subject, object) rows are inputs of the backward pass, exactly as autograd treats them.

Not differentiated here (the remaining part of f-4): the mask branch (`mask_embed`, the mask
logits, point sampling), the nine masked decoder layers, the pixel decoder and the backbone.

Checked against autograd through the reference-pinned oracle (tests/test_grad_gpu.py: 1e-4
relative on `reldec.npz`, `ppn_sep.npz` and the queries of an 800x1333 image).
"""
import math
from collections import OrderedDict

import torch

from . import hip

__all__ = ["RelationTailGrad", "HeadGrad", "PixelDecoderGrad", "BackboneGrad"]


class RelationTailGrad:
    """tape = RelationTailGrad(head); out = tape.forward(q); dq, grads = tape.backward(...)

    `q`: [B * Q, 256] fp32 on the head's device, batch-major (a plan's `pl.q`: the last
    masked-decoder layer's output BEFORE `post_norm`).  `forward` returns dict(rel [B, R, C],
    importance [B, Q, Q], importance_raw, sub / obj [B, R, nc], cls [B, Q, nc], sub_pos,
    obj_pos); the pair list is the top-k of `importance` unless `sub_pos` / `obj_pos` [B, R]
    int64 are given.  `backward(g_rel, g_importance, g_sub, g_obj, cls_detached=True)` (any subset;
    shapes of the outputs) returns (dq [B * Q, 256], {reference parameter name: gradient}).
    """

    def __init__(self, head, flat=None, base=0):
        """`flat`, `base`: keep the gradients in flat[base : base + flat_numel] of a caller-owned
        buffer (several tapes sharing one buffer: one optimizer launch, one reducer)."""
        if head.device is None or head.device.type != "cuda":
            raise RuntimeError("RelationTailGrad needs a head on an MI355X (.to('cuda:N')); "
                               "there is no CPU path")
        if head.w is None:
            head._pack()
        self.head, self.dev = head, head.device
        self.Q, self.R = head.num_obj_query, head.num_rel_query
        self.L = head.num_rel_layers
        self.ffn = head.rel_ffn
        self.scale = 1.0 / math.sqrt(32.0)
        self.t = None
        self._build_layout(head, flat, base)

    def _build_layout(self, head, flat, base):
        """Every gradient is a view of ONE flat buffer, parameters in the order their gradients are
        COMPLETED by backward() (rel_cls_embed, relation layers last to first, the relation
        embeddings, the Matrix Learner, the two PPN MLPs, then the optional class path): the
        optimizer runs over it in one launch and a data-parallel reducer can all-reduce a
        finished prefix while the rest of the backward pass is still running (train.py, dist.py)."""
        self.layout, off = OrderedDict(), 0
        self.group_end = OrderedDict()           # group name -> end offset of its last segment
        for group, names in self.param_groups(head):
            for n in names:
                shape = tuple(head._params[n].shape)
                numel = int(torch.Size(shape).numel())
                self.layout[n] = (off, shape, numel)
                off += (numel + 63) // 64 * 64
            self.group_end[group] = off
        self.flat_numel = off
        if flat is None:
            self.flat_grad = torch.zeros(off, device=self.dev, dtype=torch.float32)
        else:
            self.flat_grad = flat[base:base + off]
            assert self.flat_grad.numel() == off and base % 64 == 0
        self.grads = {n: self.flat_grad[o:o + k].view(shape)
                      for n, (o, shape, k) in self.layout.items()}

    @classmethod
    def size_of(cls, head):
        """Floats of the flat gradient buffer of this tape for `head` (segments padded to 64)."""
        return sum((int(torch.Size(tuple(head._params[n].shape)).numel()) + 63) // 64 * 64
                   for _, names in cls.param_groups(head) for n in names)

    @staticmethod
    def param_groups(head):
        """[(group, [reference parameter names])] in the order backward() completes them."""
        L = head.num_rel_layers
        groups = [("rel_cls_embed", ["rel_cls_embed.weight", "rel_cls_embed.bias"])]
        for i in reversed(range(L)):
            pre = "relation_decoder.layers.%d." % i
            names = []
            for a in ("attentions.0.attn.", "attentions.1.attn."):
                names += [pre + a + n for n in ("in_proj_weight", "in_proj_bias", "out_proj.weight",
                                                "out_proj.bias")]
            names += [pre + "norms.%d.%s" % (j, n) for j in range(3) for n in ("weight", "bias")]
            names += [pre + "ffns.0.layers.0.0.weight", pre + "ffns.0.layers.0.0.bias",
                      pre + "ffns.0.layers.1.weight", pre + "ffns.0.layers.1.bias"]
            groups.append(("relation_decoder.layers.%d" % i, names))
        groups.append(("rel_query", ["rel_query_feat.weight", "rel_query_embed.weight",
                                     "rel_query_embed2.weight"]))
        ml = "update_importance.conv_layers."
        groups.append(("update_importance", [ml + "%d.0.%s" % (j, n) for j in (2, 1, 0)
                                             for n in ("weight", "bias")]))
        for side in ("sub", "obj"):
            groups.append((side + "_query_update", ["%s_query_update.%d.%s" % (side, j, n)
                                                    for j in (4, 2, 0) for n in ("weight", "bias")]))
        groups.append(RelationTailGrad.CLS_GROUP)
        return groups

    CLS_GROUP = ("cls", ["cls_embed.weight", "cls_embed.bias", "transformer_decoder.post_norm.weight",
                         "transformer_decoder.post_norm.bias"])

    @staticmethod
    def _layer_names(pre):
        names = []
        for a in ("attentions.0.attn.", "attentions.1.attn."):
            names += [pre + a + n for n in ("in_proj_weight", "in_proj_bias", "out_proj.weight",
                                            "out_proj.bias")]
        names += [pre + "norms.%d.%s" % (j, n) for j in range(3) for n in ("weight", "bias")]
        names += [pre + "ffns.0.layers.0.0.weight", pre + "ffns.0.layers.0.0.bias",
                  pre + "ffns.0.layers.1.weight", pre + "ffns.0.layers.1.bias"]
        return names

    # ------------------------------------------------------------------ small helpers
    def _E(self, *shape):
        return torch.empty(*shape, device=self.dev, dtype=torch.float32)

    def _lin_bwd(self, dy, x, W, grads, wname, bname, row0=0, need_dx=True):
        """y = x W^T + b.  dy [M, N] (row stride free), x [M, K] contiguous, W [N, K] rows of a
        reference parameter (`row0`: their offset inside it).  Accumulates d W / d b into
        grads[wname][row0:row0+N] / grads[bname][row0:row0+N]; returns d x [M, K] or None."""
        M, K = x.shape
        N = W.shape[0]
        assert N % 4 == 0 and K % 4 == 0, (M, N, K)
        # the dW contraction runs over M: padded with zero rows to whole 32-deep k-chunks of the
        # GEMM kernels (no ragged-tail slow path; M % 4 == 0 is required anyway)
        Mp = (M + 31) // 32 * 32
        dyw = dy
        if Mp != M:
            dyw = torch.zeros(Mp, N, device=self.dev, dtype=torch.float32)
            dyw[:M].copy_(dy)
        xt = self._E(K, Mp)
        hip.transpose(x, xt)                   # (columns M .. Mp-1 zero-filled)
        gw = grads[wname][row0:row0 + N]
        tmp = self._E(N, K)
        # dW[n][k] = sum_m dy[m][n] x[m][k]: A = dy read column-major, "W" operand = x^T;
        # a long contraction over few output tiles is split (deterministic split-K + reduce)
        # (few output tiles, tens of thousands of rows to contract: 16 K-slices on the 64x64 tile
        # kernel instead of the 32x32 skinny kernel's 64 workgroups -- 134 -> ~30 us at 21 950 rows)
        kw = dict(scratch=self._E(16 * N * K), force="tile64", ksplit=16) if Mp >= 2048 else {}
        hip.gemm(dyw, xt, tmp, M=N, N=K, K=Mp, lda=dyw.stride(0), ldw=Mp, ldc=K, colmajor=True, **kw)
        hip.add_periodic(gw, tmp, gw)
        if bname is not None:
            hip.colsum(dy, grads[bname][row0:row0 + N], accumulate=True)
        if not need_dx:
            return None
        Wt = self._E(K, N)
        hip.transpose(W, Wt)
        dx = self._E(M, K)
        hip.gemm(dy, Wt, dx, M=M, N=K, K=N, lda=dy.stride(0), ldw=N, ldc=K)
        return dx

    def _ln_bwd(self, dy, x, prefix, grads):
        """LayerNorm(256) backward from its saved input; accumulates d weight / d bias."""
        w = self.head.w
        dx, gx = self._E(*x.shape), self._E(*x.shape)
        hip.layernorm256_bwd(dy, x, w[prefix + "weight"], dx, gx)
        hip.colsum(gx, grads[prefix + "weight"], accumulate=True)
        hip.colsum(dy, grads[prefix + "bias"], accumulate=True)
        return dx

    def _acc(self, a, b):
        hip.add_periodic(a, b, a)

    # ------------------------------------------------------------------ forward with a tape
    @torch.no_grad()
    @hip.on_device
    def forward(self, q, sub_pos=None, obj_pos=None):
        head, w, E = self.head, self.head.w, self._E
        Q, R = self.Q, self.R
        if not (q.is_cuda and q.dtype == torch.float32 and q.dim() == 2 and q.shape[1] == 256
                and q.is_contiguous() and q.shape[0] % Q == 0):
            raise RuntimeError("q must be a contiguous [B * Q, 256] fp32 device tensor")
        B = q.shape[0] // Q
        t = self.t = dict(B=B, q=q)
        # ---- post_norm + cls_embed (pairnet_head.py:236-238) ----
        nc = head.num_classes + 1
        t["qn"] = E(B * Q, 256)
        hip.layernorm(q, w["transformer_decoder.post_norm.weight"],
                      w["transformer_decoder.post_norm.bias"], t["qn"])
        cls = E(B, Q, nc)
        hip.linear(t["qn"], w["cls_embed.weight"], w["cls_embed.bias"], cls.view(B * Q, nc))
        # ---- Pair Proposal Network (:322-340) ----
        for side in ("sub", "obj"):
            mlp = side + "_query_update"
            h1, h2, e = E(B * Q, 256), E(B * Q, 256), E(B * Q, 256)
            hip.linear(q, w[mlp + ".0.weight"], w[mlp + ".0.bias"], h1, relu=True)
            hip.linear(h1, w[mlp + ".2.weight"], w[mlp + ".2.bias"], h2, relu=True)
            hip.linear(h2, w[mlp + ".4.weight"], w[mlp + ".4.bias"], e)
            t[side] = (h1, h2, e)
        ml = "update_importance.conv_layers."
        raw, c1, c2, imp = E(B, Q, Q), E(B, Q, Q, 64), E(B, Q, Q, 64), E(B, Q, Q)
        hip.ppn_front(t["sub"][2], t["obj"][2], w[ml + "0.0.weight"], w[ml + "0.0.bias"], raw, c1,
                      B, Q)
        hip.conv2d_ex(c1, w[ml + "1.0.weight"], w[ml + "1.0.bias"], None, c2, B, Q, Q, 64, 64, 7,
                      7, 1, 3, relu=True)
        hip.mlearner_last(c2, w[ml + "2.0.weight"], w[ml + "2.0.bias"], imp, B, Q)
        t.update(raw=raw, c1=c1, c2=c2)
        i64 = lambda *s: torch.empty(*s, device=self.dev, dtype=torch.int64)
        pair_idx = i64(B, 2 * R)
        if sub_pos is None:
            topk, sub_pos, obj_pos = i64(B, R), i64(B, R), i64(B, R)
            hip.topk_pairs(imp, topk, sub_pos, obj_pos, B, Q, R, pair=pair_idx)
        else:
            sub_pos = sub_pos.to(self.dev, torch.int64).contiguous()
            obj_pos = obj_pos.to(self.dev, torch.int64).contiguous()
            pair_idx[:, :R].copy_(sub_pos)
            pair_idx[:, R:].copy_(obj_pos)
        t.update(sub_pos=sub_pos, obj_pos=obj_pos, pair_idx=pair_idx)
        # ---- pair features (:342-351) ----
        pair = E(B * 2 * R, 256)
        hip.gather_rows(q, pair_idx, pair, B, Q, 2 * R, 256)
        rel = self._relation_forward(pair, B)
        sub, obj = E(B, R, nc), E(B, R, nc)
        hip.gather_rows(cls, sub_pos, sub, B, Q, R, nc)
        hip.gather_rows(cls, obj_pos, obj, B, Q, R, nc)
        return dict(rel=rel, importance=imp, importance_raw=raw, sub=sub, obj=obj, cls=cls,
                    sub_pos=sub_pos, obj_pos=obj_pos)

    def _layer_fwd(self, pre, x, qpos, K, V, B, nq, nk, ffn, scr, bits=None, rowall=None):
        """One post-norm decoder layer (facebook_detr.py:378-432; order cross-attention, norm,
        self-attention, norm, FFN, norm) over given key / value projections K, V (2-D views
        [B * nk, 256], free row strides) with every intermediate kept; returns (tape, output)."""
        w, E = self.head.w, self._E
        M = B * nq
        ac, as_ = pre + "attentions.0.attn.", pre + "attentions.1.attn."
        Wc, bc = w[ac + "in_proj_weight"], w[ac + "in_proj_bias"]
        Ws, bs = w[as_ + "in_proj_weight"], w[as_ + "in_proj_bias"]
        s = dict(x_in=x, K=K, V=V, bits=bits, rowall=rowall)
        # cross-attention
        s["xp"] = E(M, 256)
        hip.add_periodic(x, qpos, s["xp"])
        s["Qc"] = E(M, 256)
        hip.linear(s["xp"], Wc[:256], bc[:256], s["Qc"])
        s["att_c"] = E(M, 256)
        hip.attention(s["Qc"], 256, K, K.stride(0), V, V.stride(0), bits, rowall, s["att_c"], 256,
                      scr, B, nq, nk, self.scale)
        s["y1"] = E(M, 256)
        hip.linear(s["att_c"], w[ac + "out_proj.weight"], w[ac + "out_proj.bias"], s["y1"], res=x)
        s["x1"] = E(M, 256)
        hip.layernorm(s["y1"], w[pre + "norms.0.weight"], w[pre + "norms.0.bias"], s["x1"])
        # self-attention
        s["x1p"] = E(M, 256)
        hip.add_periodic(s["x1"], qpos, s["x1p"])
        s["QKVs"] = E(M, 768)                               # columns [Q | K | V]
        hip.linear(s["x1p"], Ws[:512], bs[:512], s["QKVs"][:, :512])
        hip.linear(s["x1"], Ws[512:], bs[512:], s["QKVs"][:, 512:])
        s["att_s"] = E(M, 256)
        hip.attention(s["QKVs"], 768, s["QKVs"][:, 256:], 768, s["QKVs"][:, 512:], 768, None,
                      None, s["att_s"], 256, scr, B, nq, nq, self.scale)
        s["y2"] = E(M, 256)
        hip.linear(s["att_s"], w[as_ + "out_proj.weight"], w[as_ + "out_proj.bias"], s["y2"],
                   res=s["x1"])
        s["x2"] = E(M, 256)
        hip.layernorm(s["y2"], w[pre + "norms.1.weight"], w[pre + "norms.1.bias"], s["x2"])
        # FFN
        s["h"] = E(M, ffn)
        hip.linear(s["x2"], w[pre + "ffns.0.layers.0.0.weight"],
                   w[pre + "ffns.0.layers.0.0.bias"], s["h"], relu=True)
        s["y3"] = E(M, 256)
        hip.linear(s["h"], w[pre + "ffns.0.layers.1.weight"], w[pre + "ffns.0.layers.1.bias"],
                   s["y3"], res=s["x2"])
        out = E(M, 256)
        hip.layernorm(s["y3"], w[pre + "norms.2.weight"], w[pre + "norms.2.bias"], out)
        return s, out

    def _layer_bwd(self, pre, s, dx, grads, B, nq, nk, scr, dqpos_rows):
        """Backward of `_layer_fwd`: accumulates the layer's parameter gradients and d (x +
        query_pos) (all three uses) into `dqpos_rows`; returns (d input, d K, d V)."""
        w, E = self.head.w, self._E
        M = B * nq
        ac, as_ = pre + "attentions.0.attn.", pre + "attentions.1.attn."
        Wc, Ws = w[ac + "in_proj_weight"], w[as_ + "in_proj_weight"]
        # norm2 <- FFN
        dy3 = self._ln_bwd(dx, s["y3"], pre + "norms.2.", grads)
        dh = self._lin_bwd(dy3, s["h"], w[pre + "ffns.0.layers.1.weight"], grads,
                           pre + "ffns.0.layers.1.weight", pre + "ffns.0.layers.1.bias")
        hip.relu_bwd(dh, s["h"], dh)
        dx2 = self._lin_bwd(dh, s["x2"], w[pre + "ffns.0.layers.0.0.weight"], grads,
                            pre + "ffns.0.layers.0.0.weight", pre + "ffns.0.layers.0.0.bias")
        self._acc(dx2, dy3)                                   # the FFN's identity shortcut
        # norm1 <- self-attention
        dy2 = self._ln_bwd(dx2, s["y2"], pre + "norms.1.", grads)
        datt = self._lin_bwd(dy2, s["att_s"], w[as_ + "out_proj.weight"], grads,
                             as_ + "out_proj.weight", as_ + "out_proj.bias")
        dQKV = E(M, 768)
        hip.mha_bwd(s["QKVs"], s["QKVs"][:, 256:], s["QKVs"][:, 512:], datt, dQKV,
                    dQKV[:, 256:], dQKV[:, 512:], scr, B, nq, nq, self.scale)
        dx1p = self._lin_bwd(dQKV[:, :512], s["x1p"], Ws[:512], grads, as_ + "in_proj_weight",
                             as_ + "in_proj_bias", row0=0)
        dx1 = self._lin_bwd(dQKV[:, 512:], s["x1"], Ws[512:], grads, as_ + "in_proj_weight",
                            as_ + "in_proj_bias", row0=512)
        self._acc(dx1, dx1p)
        self._acc(dqpos_rows, dx1p)
        self._acc(dx1, dy2)                                   # identity shortcut
        # norm0 <- cross-attention
        dy1 = self._ln_bwd(dx1, s["y1"], pre + "norms.0.", grads)
        datt = self._lin_bwd(dy1, s["att_c"], w[ac + "out_proj.weight"], grads,
                             ac + "out_proj.weight", ac + "out_proj.bias")
        dQc, dK, dV = E(M, 256), E(B * nk, 256), E(B * nk, 256)
        hip.mha_bwd(s["Qc"], s["K"], s["V"], datt, dQc, dK, dV, scr, B, nq, nk, self.scale,
                    bits=s["bits"], rowall=s["rowall"])
        dxp = self._lin_bwd(dQc, s["xp"], Wc[:256], grads, ac + "in_proj_weight",
                            ac + "in_proj_bias", row0=0)
        self._acc(dqpos_rows, dxp)
        self._acc(dxp, dy1)                                   # identity shortcut
        return dxp, dK, dV

    def _relation_forward(self, pair, B):
        """The six Relation Fusion layers over `pair` [B * 2R, 256] with every intermediate kept."""
        head, w, E = self.head, self.head.w, self._E
        R, t = self.R, self.t
        M, Mk = B * R, B * 2 * R
        rpos, ppos = w["rel_query_embed.weight"], w["rel_query_embed2.weight"]
        t["pair"] = pair
        pairp = E(Mk, 256)                       # pair + key_pos (the keys' operand)
        hip.add_periodic(pair, ppos, pairp)
        t["pairp"] = pairp
        x = head._const(B)["r0"]                 # rel_query_feat repeated over the batch
        layers = []
        scr = E(max(hip.attn_scratch_floats(B, R, 2 * R), hip.attn_scratch_floats(B, R, R)))
        for i in range(self.L):
            pre = "relation_decoder.layers.%d." % i
            ac = pre + "attentions.0.attn."
            Wc, bc = w[ac + "in_proj_weight"], w[ac + "in_proj_bias"]
            KV = E(Mk, 512)                                     # columns [K | V]
            hip.linear(pairp, Wc[256:512], bc[256:512], KV[:, :256])
            hip.linear(pair, Wc[512:], bc[512:], KV[:, 256:])
            s, x = self._layer_fwd(pre, x, rpos, KV[:, :256], KV[:, 256:], B, R, 2 * R, self.ffn,
                                   scr)
            layers.append(s)
        t["layers"], t["r_out"] = layers, x
        C = w["rel_cls_embed.weight"].shape[0]
        rel = E(B, R, C)
        hip.linear(x, w["rel_cls_embed.weight"], w["rel_cls_embed.bias"], rel.view(M, C))
        return rel

    # ------------------------------------------------------------------ backward
    def _zero_grads(self):
        self.flat_grad.zero_()
        return self.grads

    @torch.no_grad()
    @hip.on_device
    def backward(self, g_rel=None, g_importance=None, g_sub=None, g_obj=None, cls_detached=True,
                 on_ready=None):
        """`cls_detached` (default: the reference's graph): pairnet_head.py:380-390 gathers the
        subject / object class logits from `cls_pred.clone().detach()`, so `loss_sub_cls` and
        `loss_obj_cls` reach no parameter -- `g_sub` / `g_obj` are accepted and, like autograd
        does there, contribute nothing.  With `cls_detached=False` they are propagated through
        the gathers, `cls_embed` and `post_norm` (the derivative of the un-detached expression).
        The gradients are views of `self.flat_grad` (valid until the next backward);
        `on_ready(end)`: called whenever flat_grad[:end] has become final (a reducer's hook)."""
        dq, grads = self._backward_tail(g_rel, g_importance, g_sub, g_obj, cls_detached, on_ready)
        if on_ready is not None:
            on_ready(self.flat_numel)
        return dq, grads

    def _backward_tail(self, g_rel, g_importance, g_sub, g_obj, cls_detached, on_ready):
        if self.t is None:
            raise RuntimeError("backward() needs a forward() first")
        t, B, Q, R = self.t, self.t["B"], self.Q, self.R
        grads = self._zero_grads()
        dq = torch.zeros(B * Q, 256, device=self.dev, dtype=torch.float32)
        prep = lambda g: g.to(self.dev, torch.float32).contiguous()
        ready = on_ready if on_ready is not None else (lambda end: None)
        if g_rel is not None:
            # ... and the gather of the pair features back onto the query rows (:342-351)
            dpair = self._relation_backward(prep(g_rel), grads, on_ready)
            hip.scatter_rows_add(dpair, t["pair_idx"], dq, B, Q, 2 * R, 256, accumulate=True)
        ready(self.group_end["rel_query"])
        if g_importance is not None:
            self._ppn_backward(prep(g_importance), grads, dq)
        ready(self.group_end["obj_query_update"])
        if not cls_detached and (g_sub is not None or g_obj is not None):
            self._cls_backward(g_sub, g_obj, grads, dq)
        return dq, grads

    @torch.no_grad()
    @hip.on_device
    def relation_forward(self, pair):
        """The Relation Fusion decoder alone: pair features [B * 2R, 256] (per image the R
        subject rows, then the R object rows) -> relation logits [B, R, C], taped."""
        B = pair.shape[0] // (2 * self.R)
        self.t = dict(B=B, q=None)
        return self._relation_forward(pair.contiguous(), B)

    @torch.no_grad()
    @hip.on_device
    def relation_backward(self, g_rel):
        """-> (d pair features [B * 2R, 256], {parameter name: gradient}) of the last
        `relation_forward` / `forward`."""
        grads = self._zero_grads()
        return self._relation_backward(g_rel.to(self.dev, torch.float32).contiguous(), grads), grads

    def _relation_backward(self, g_rel, grads, on_ready=None):
        w, E, t = self.head.w, self._E, self.t
        B, Q, R = t["B"], self.Q, self.R
        M, Mk = B * R, B * 2 * R
        C = g_rel.shape[-1]
        ready = on_ready if on_ready is not None else (lambda end: None)
        dx = self._lin_bwd(g_rel.view(M, C), t["r_out"], w["rel_cls_embed.weight"], grads,
                           "rel_cls_embed.weight", "rel_cls_embed.bias")
        ready(self.group_end["rel_cls_embed"])
        zeros = lambda *s_: torch.zeros(*s_, device=self.dev, dtype=torch.float32)
        dpair, dpairp = zeros(Mk, 256), zeros(Mk, 256)       # d pair, d (pair + key_pos)
        drpos_rows = zeros(M, 256)                            # d (x + query_pos), all uses
        scr = E(max(hip.mha_bwd_scratch_floats(B, R, 2 * R), hip.mha_bwd_scratch_floats(B, R, R)))
        for i in reversed(range(self.L)):
            pre = "relation_decoder.layers.%d." % i
            ac = pre + "attentions.0.attn."
            Wc = w[ac + "in_proj_weight"]
            dx, dK, dV = self._layer_bwd(pre, t["layers"][i], dx, grads, B, R, 2 * R, scr,
                                         drpos_rows)
            dkp = self._lin_bwd(dK, t["pairp"], Wc[256:512], grads, ac + "in_proj_weight",
                                ac + "in_proj_bias", row0=256)
            self._acc(dpairp, dkp)
            dv = self._lin_bwd(dV, t["pair"], Wc[512:], grads, ac + "in_proj_weight",
                               ac + "in_proj_bias", row0=512)
            self._acc(dpair, dv)
            ready(self.group_end["relation_decoder.layers.%d" % i])
        # layer 0's input is rel_query_feat repeated over the batch
        hip.batch_sum(dx, grads["rel_query_feat.weight"], B)
        hip.batch_sum(drpos_rows, grads["rel_query_embed.weight"], B)
        hip.batch_sum(dpairp, grads["rel_query_embed2.weight"], B)
        self._acc(dpair, dpairp)
        return dpair

    def _mlp3_bwd(self, d_out, x, saved, prefix, grads):
        """Linear-ReLU-Linear-ReLU-Linear backward; saved = (h1, h2, out)."""
        w = self.head.w
        h1, h2, _ = saved
        d2 = self._lin_bwd(d_out, h2, w[prefix + ".4.weight"], grads, prefix + ".4.weight",
                           prefix + ".4.bias")
        hip.relu_bwd(d2, h2, d2)
        d1 = self._lin_bwd(d2, h1, w[prefix + ".2.weight"], grads, prefix + ".2.weight",
                           prefix + ".2.bias")
        hip.relu_bwd(d1, h1, d1)
        return self._lin_bwd(d1, x, w[prefix + ".0.weight"], grads, prefix + ".0.weight",
                             prefix + ".0.bias")

    def _ppn_backward(self, g_imp, grads, dq):
        w, E, t = self.head.w, self._E, self.t
        B, Q = t["B"], self.Q
        ml = "update_importance.conv_layers."
        c1, c2, raw = t["c1"], t["c2"], t["raw"]
        npix = B * Q * Q
        # ---- last layer (64 -> 1) ----
        part = E(B * Q, 49 * 64)
        hip.tapcorr1(c2, g_imp, part, B, Q, -1)
        dw3 = E(49 * 64)                                          # [tap][ci]
        hip.colsum(part, dw3)
        db3 = E(1)
        hip.colsum(g_imp.view(npix, 1), db3)
        dc2 = E(B, Q, Q, 64)
        hip.mlearner_last_bwd_data(g_imp, w[ml + "2.0.weight"], c2, dc2, B, Q)
        # ---- middle layer (64 -> 64) ----
        rows_per = 10
        chunks = B * ((Q + rows_per - 1) // rows_per)
        part2 = E(chunks, 64 * 49 * 64)
        hip.conv_wgrad(dc2, c1, part2, B, Q, Q, Q, Q, 64, 64, 7, 1, 3, rows_per)
        dw2 = E(64 * 49 * 64)                                     # [co][tap][ci]
        hip.colsum(part2, dw2)
        db2 = E(64)
        hip.colsum(dc2.view(npix, 64), db2)
        w2b = E(64 * 49 * 64)
        hip.conv_weight_bwd_layout(w[ml + "1.0.weight"], w2b, 64, 49, 64)
        dc1 = E(B, Q, Q, 64)
        hip.conv2d_ex(dc2, w2b.view(64, 49 * 64), None, None, dc1, B, Q, Q, 64, 64, 7, 7, 1, 3)
        hip.relu_bwd(dc1, c1, dc1)
        # ---- first layer (1 -> 64) ----
        hip.tapcorr1(dc1, raw, part, B, Q, +1)
        dw1t = E(49 * 64)                                         # [tap][co]
        hip.colsum(part, dw1t)
        db1 = E(64)
        hip.colsum(dc1.view(npix, 64), db1)
        w1b = E(49 * 64)
        hip.conv_weight_bwd_layout(w[ml + "0.0.weight"], w1b, 64, 49, 1)    # [1][49 flipped][64]
        draw = E(B, Q, Q)
        zero = torch.zeros(1, device=self.dev, dtype=torch.float32)
        hip.mlearner_last(dc1, w1b.view(49, 64), zero, draw, B, Q)
        # parameter gradients in the reference's layouts (cnn_factory.py: Conv2d weights [Co][Ci][7][7])
        # (copies that only re-order: [tap][ci] -> [1][ci][7][7], [co][tap][ci] -> [co][ci][7][7], ...)
        grads[ml + "2.0.weight"].copy_(dw3.view(49, 64).t().reshape(1, 64, 7, 7))
        grads[ml + "2.0.bias"].copy_(db3)
        grads[ml + "1.0.weight"].copy_(dw2.view(64, 7, 7, 64).permute(0, 3, 1, 2))
        grads[ml + "1.0.bias"].copy_(db2)
        grads[ml + "0.0.weight"].copy_(dw1t.view(49, 64).t().reshape(64, 1, 7, 7))
        grads[ml + "0.0.bias"].copy_(db1)
        # ---- cosine block + F.normalize ----
        s_e, o_e = t["sub"][2], t["obj"][2]
        s_hat, o_hat = E(B * Q, 256), E(B * Q, 256)
        hip.l2normalize(s_e, s_hat)
        hip.l2normalize(o_e, o_hat)
        ds, do = E(B * Q, 256), E(B * Q, 256)
        hip.cosine_bwd(draw, s_e, o_hat, ds, B, Q, False)
        hip.cosine_bwd(draw, o_e, s_hat, do, B, Q, True)
        self._acc(dq, self._mlp3_bwd(ds, t["q"], t["sub"], "sub_query_update", grads))
        self._acc(dq, self._mlp3_bwd(do, t["q"], t["obj"], "obj_query_update", grads))

    def _cls_backward(self, g_sub, g_obj, grads, dq):
        w, t = self.head.w, self.t
        B, Q, R = t["B"], self.Q, self.R
        nc = self.head.num_classes + 1
        ncp = (nc + 3) // 4 * 4           # the GEMMs contract over multiples of 4: zero columns
        zeros = lambda *s: torch.zeros(*s, device=self.dev, dtype=torch.float32)
        dcls = zeros(B * Q, ncp)
        for g, pos in ((g_sub, t["sub_pos"]), (g_obj, t["obj_pos"])):
            if g is not None:
                g = g.to(self.dev, torch.float32).contiguous().view(B * R, nc)
                hip.scatter_rows_add(g, pos, dcls, B, Q, R, nc, accumulate=True)
        Wp = zeros(ncp, 256)
        Wp[:nc].copy_(w["cls_embed.weight"])
        gp = {"W": zeros(ncp, 256), "b": zeros(ncp)}
        dqn = self._lin_bwd(dcls, t["qn"], Wp, gp, "W", "b")
        self._acc(grads["cls_embed.weight"], gp["W"][:nc])
        self._acc(grads["cls_embed.bias"], gp["b"][:nc])
        self._acc(dq, self._ln_bwd(dqn, t["q"], "transformer_decoder.post_norm.", grads))


class HeadGrad(RelationTailGrad):
    """RelationTailGrad + the nine masked-attention decoder layers in front of it
    (pairnet_head.py:289-320; the reference's `transformer_decoder`, trained at lr_mult 0.1):
    everything of CrossHead2 that the reference's loss reaches behind the pixel decoder.

        tape = HeadGrad(head); head.forward(feats, metas); pl = head._last_plan
        out = tape.forward_from_plan(pl)               # re-runs the query chain with a tape
        dmem, grads = tape.backward(g_rel, g_importance)

    `forward_from_plan` needs a plan whose stage A has run (`head.forward`): the memory tokens
    `pl.X`, the K / V projections of the nine layers, the stencil rows of the mask feature.  The
    boolean attention masks (:244-256, `detach()`ed in the reference) are recomputed per layer by
    the product's own fused kernel and kept as packed bits.  `backward` returns the gradient with
    respect to the pixel decoder's memory tokens [B, sum_l N_l, 256] (what
    `PixelDecoderGrad.backward` takes) and the parameter gradients incl. `query_feat`,
    `query_embed`, `level_embed` and the cross-attentions' K / V projection rows."""

    @staticmethod
    def param_groups(head):
        groups = [g for g in RelationTailGrad.param_groups(head) if g[0] != "cls"]
        for i in reversed(range(head.num_dec_layers)):
            groups.append(("transformer_decoder.layers.%d" % i,
                           RelationTailGrad._layer_names("transformer_decoder.layers.%d." % i)))
        groups.append(("query", ["query_feat.weight", "query_embed.weight", "level_embed.weight"]))
        groups.append(RelationTailGrad.CLS_GROUP)
        return groups

    @torch.no_grad()
    @hip.on_device
    def forward_from_plan(self, pl, sub_pos=None, obj_pos=None):
        head, w, E = self.head, self.head.w, self._E
        B, Q = pl.B, self.Q
        full = head.exact_mask_order == "full"
        qpos = w["query_embed.weight"]
        x = pl.q0
        if full:
            head._head_embed(pl.q0, pl, False, True)
        scr = E(max([hip.attn_scratch_floats(B, Q, n) for n in pl.N] +
                    [hip.attn_scratch_floats(B, Q, Q)]))
        layers = []
        for i in range(head.num_dec_layers):
            l = i % 3
            head._attn_mask(pl, l, None, me=pl.me0 if (i == 0 and not full) else None)
            nw = (pl.N[l] + 31) // 32
            bits, rowall = pl.bits[:B * Q * nw].clone(), pl.rowall.clone()
            s, x = self._layer_fwd("transformer_decoder.layers.%d." % i, x, qpos,
                                   pl.Kp[i].view(B * pl.N[l], 256), pl.Vp[i].view(B * pl.N[l], 256),
                                   B, Q, pl.N[l], head.dec_ffn, scr, bits, rowall)
            s["level"] = l
            layers.append(s)
            if i + 1 < head.num_dec_layers:      # post_norm + mask_embed -> the next layer's mask
                head._head_embed(x, pl, False, full)
        self.dt = dict(layers=layers, pl=pl, q_out=x)
        return self.forward(x, sub_pos, obj_pos)

    @torch.no_grad()
    @hip.on_device
    def backward(self, g_rel=None, g_importance=None, g_sub=None, g_obj=None, cls_detached=True,
                 on_ready=None):
        """-> (d memory tokens [B, SN, 256], {parameter name: gradient}); `self.dq_out` keeps the
        gradient w.r.t. the decoder's output queries."""
        dq, grads = self._backward_tail(g_rel, g_importance, g_sub, g_obj, cls_detached, on_ready)
        self.dq_out = dq
        head, w, E = self.head, self.head.w, self._E
        ready = on_ready if on_ready is not None else (lambda end: None)
        pl, layers = self.dt["pl"], self.dt["layers"]
        B, Q = pl.B, self.Q
        zeros = lambda *s_: torch.zeros(*s_, device=self.dev, dtype=torch.float32)
        dmem = zeros(B, pl.SN, 256)
        dqpos_rows = zeros(B * Q, 256)
        scr = E(max([hip.mha_bwd_scratch_floats(B, Q, n) for n in pl.N] +
                    [hip.mha_bwd_scratch_floats(B, Q, Q)]))
        le, dle = w["level_embed.weight"], grads["level_embed.weight"]
        dx = dq.clone()
        for i in reversed(range(head.num_dec_layers)):
            pre = "transformer_decoder.layers.%d." % i
            ac = pre + "attentions.0.attn."
            Wc = w[ac + "in_proj_weight"]
            s = layers[i]
            l, N = s["level"], pl.N[s["level"]]
            dx, dK, dV = self._layer_bwd(pre, s, dx, grads, B, Q, N, scr, dqpos_rows)
            # K = (mem_l + level_embed_l + pe_l) Wk^T + bk, V = (mem_l + level_embed_l) Wv^T + bv
            # (pairnet_head.py:278-287, :302-312), image by image: a level's tokens of one image
            # are contiguous rows of pl.X
            for b in range(B):
                mem = pl.X[b, pl.start[l]:pl.start[l] + N]
                memk, memv = E(N, 256), E(N, 256)
                hip.add_periodic(mem, pl.dec_kpos[l], memk)
                hip.add_periodic(mem, le[l:l + 1], memv)
                dmk = self._lin_bwd(dK[b * N:(b + 1) * N], memk, Wc[256:512], grads,
                                    ac + "in_proj_weight", ac + "in_proj_bias", row0=256)
                dmv = self._lin_bwd(dV[b * N:(b + 1) * N], memv, Wc[512:], grads,
                                    ac + "in_proj_weight", ac + "in_proj_bias", row0=512)
                self._acc(dmk, dmv)
                dm = dmem[b, pl.start[l]:pl.start[l] + N]
                self._acc(dm, dmk)
                hip.colsum(dmk, dle[l], accumulate=True)      # level_embed_l feeds K and V
            ready(self.group_end["transformer_decoder.layers.%d" % i])
        hip.batch_sum(dx, grads["query_feat.weight"], B)
        hip.batch_sum(dqpos_rows, grads["query_embed.weight"], B)
        ready(self.group_end["query"])
        ready(self.flat_numel)
        return dmem, grads


class PixelDecoderGrad(RelationTailGrad):
    """The pixel decoder's path from the backbone features to the memory tokens the masked decoder
    attends to (MSDeformAttnPixelDecoder behind pairnet_head.py:262: three 1x1 input convolutions +
    GroupNorm, six deformable-attention encoder layers), taped and differentiated:

        tape = PixelDecoderGrad(head)
        mem = tape.forward(feats)                     # [B, sum_l N_l, 256], levels coarsest first
        dfeats, grads = tape.backward(dmem)           # dmem e.g. from HeadGrad.backward

    `dfeats[l]` is the gradient w.r.t. the level-l feature map (`feats[3 - l]`: C5, C4, C3) as a
    contiguous [B, C_l, h, w] tensor -- where a backbone's backward would continue.  The mask-feature
    branch (lateral / output convolutions, `mask_feature`) is not here: the reference's loss does not
    reach it (the masks are detached, pairnet_head.py:256, :391-398).  The sampling operator is
    differentiated in mmcv's own operand set (`pn_msda_loc_f32` forward, `pn_msda_bwd_f32`: atomics
    on grad_value, like mmcv's), its operands come from `pn_token_sampling_f32` and go back through
    `pn_msda_offaw_bwd_f32`; the forward runs on the exact-fp32 MFMA kernels whatever
    `head.gemm_arithmetic` says."""

    PD = "pixel_decoder."

    @staticmethod
    def param_groups(head):
        pd = PixelDecoderGrad.PD
        groups = []
        for i in reversed(range(head.num_enc_layers)):
            p = pd + "encoder.layers.%d." % i
            a = p + "attentions.0."
            names = [a + n + s for n in ("value_proj", "sampling_offsets", "attention_weights",
                                         "output_proj") for s in (".weight", ".bias")]
            names += [p + "norms.%d.%s" % (j, n) for j in range(2) for n in ("weight", "bias")]
            names += [p + "ffns.0.layers.0.0.weight", p + "ffns.0.layers.0.0.bias",
                      p + "ffns.0.layers.1.weight", p + "ffns.0.layers.1.bias"]
            groups.append(("encoder.layers.%d" % i, names))
        groups.append(("level_encoding", [pd + "level_encoding.weight"]))
        for l in range(3):
            groups.append(("input_convs.%d" % l,
                           [pd + "input_convs.%d.%s" % (l, n)
                            for n in ("conv.weight", "conv.bias", "gn.weight", "gn.bias")]))
        return groups

    @staticmethod
    def _rows(fb):
        """One image's feature map [C, h, w] as pixel rows [h*w, C] (a view for channels-last
        memory, a transpose pass for NCHW)."""
        C, h, w = fb.shape
        if fb.stride(0) == 1:
            return fb.permute(1, 2, 0).reshape(h * w, C)
        out = torch.empty(h * w, C, device=fb.device, dtype=torch.float32)
        hip.transpose(fb.reshape(C, h * w), out)
        return out

    @torch.no_grad()
    @hip.on_device
    def forward(self, feats):
        head, w, E, pd = self.head, self.head.w, self._E, self.PD
        B = feats[0].shape[0]
        shapes = [tuple(feats[3 - l].shape[-2:]) for l in range(3)]
        N = [h * wd for h, wd in shapes]
        start = [0, N[0], N[0] + N[1]]
        SN, M = sum(N), B * sum(N)
        enc_pos = head._position_tables(shapes)[0]                  # sine pe + level_encoding
        t = self.t = dict(B=B, shapes=shapes, N=N, start=start, SN=SN, convs=[], rows=[])
        X = E(B, SN, 256)
        G = head.gn_groups
        part = torch.empty(B * hip.groupnorm_nblk(max(N)) * G * 2, device=self.dev,
                           dtype=torch.float64)
        for l in range(3):
            f = feats[3 - l]
            conv = E(B, N[l], 256)
            rows = [self._rows(f[b]) for b in range(B)]
            for b in range(B):
                hip.linear(rows[b], w[pd + "input_convs.%d.conv.weight" % l],
                           w[pd + "input_convs.%d.conv.bias" % l], conv[b])
            hip.groupnorm_nhwc(conv, w[pd + "input_convs.%d.gn.weight" % l],
                               w[pd + "input_convs.%d.gn.bias" % l], X[:, start[l]:], part, B, N[l],
                               G, False, N[l] * 256, SN * 256)
            t["convs"].append(conv)
            t["rows"].append(rows)
        t["shapes_dev"] = torch.tensor(shapes, dtype=torch.int64, device=self.dev)
        t["lsi_dev"] = torch.tensor(start, dtype=torch.int64, device=self.dev)
        ones = torch.ones(B, 3, 2, device=self.dev, dtype=torch.float32)     # unpadded batch
        x = X.view(M, 256)
        layers = []
        F = head.enc_ffn
        for i in range(head.num_enc_layers):
            p = pd + "encoder.layers.%d." % i
            a = p + "attentions.0."
            s = dict(x_in=x)
            s["xq"] = E(M, 256)
            hip.add_periodic(x, enc_pos, s["xq"])
            s["value"] = E(M, 256)
            hip.linear(x, w[a + "value_proj.weight"], w[a + "value_proj.bias"], s["value"])
            offaw = E(M, 288)
            hip.linear(s["xq"], w[a + "sampling_offsets.weight"], w[a + "sampling_offsets.bias"],
                       offaw[:, :192])
            hip.linear(s["xq"], w[a + "attention_weights.weight"], w[a + "attention_weights.bias"],
                       offaw[:, 192:])
            s["loc"], s["aw"] = E(M * 8 * 3 * 4 * 2), E(M * 8 * 3 * 4)
            hip.token_sampling(offaw, 288, ones, s["loc"], s["aw"], B, shapes)
            s["S"] = E(M, 256)
            hip.msda_loc(s["value"], 256, t["shapes_dev"], t["lsi_dev"], s["loc"], s["aw"], s["S"],
                         B, SN, SN, 3)
            s["y1"] = E(M, 256)
            hip.linear(s["S"], w[a + "output_proj.weight"], w[a + "output_proj.bias"], s["y1"],
                       res=x)
            s["x1"] = E(M, 256)
            hip.layernorm(s["y1"], w[p + "norms.0.weight"], w[p + "norms.0.bias"], s["x1"])
            s["h"] = E(M, F)
            hip.linear(s["x1"], w[p + "ffns.0.layers.0.0.weight"], w[p + "ffns.0.layers.0.0.bias"],
                       s["h"], relu=True)
            s["y2"] = E(M, 256)
            hip.linear(s["h"], w[p + "ffns.0.layers.1.weight"], w[p + "ffns.0.layers.1.bias"],
                       s["y2"], res=s["x1"])
            x = E(M, 256)
            hip.layernorm(s["y2"], w[p + "norms.1.weight"], w[p + "norms.1.bias"], x)
            layers.append(s)
        t["layers"] = layers
        return x.view(B, SN, 256)

    @torch.no_grad()
    @hip.on_device
    def backward(self, dmem, on_ready=None):
        if self.t is None or "layers" not in self.t:
            raise RuntimeError("backward() needs a forward() first")
        head, w, E, pd, t = self.head, self.head.w, self._E, self.PD, self.t
        B, shapes, N, start, SN = t["B"], t["shapes"], t["N"], t["start"], t["SN"]
        M = B * SN
        ready = on_ready if on_ready is not None else (lambda end: None)
        grads = self._zero_grads()
        zeros = lambda *s_: torch.zeros(*s_, device=self.dev, dtype=torch.float32)
        dx = dmem.to(self.dev, torch.float32).contiguous().view(M, 256).clone()
        dpos_rows = zeros(M, 256)
        for i in reversed(range(head.num_enc_layers)):
            p = pd + "encoder.layers.%d." % i
            a = p + "attentions.0."
            s = t["layers"][i]
            dy2 = self._ln_bwd(dx, s["y2"], p + "norms.1.", grads)
            dh = self._lin_bwd(dy2, s["h"], w[p + "ffns.0.layers.1.weight"], grads,
                               p + "ffns.0.layers.1.weight", p + "ffns.0.layers.1.bias")
            hip.relu_bwd(dh, s["h"], dh)
            dx1 = self._lin_bwd(dh, s["x1"], w[p + "ffns.0.layers.0.0.weight"], grads,
                                p + "ffns.0.layers.0.0.weight", p + "ffns.0.layers.0.0.bias")
            del dh
            self._acc(dx1, dy2)
            dxn = self._ln_bwd(dx1, s["y1"], p + "norms.0.", grads)        # = dy1: the shortcut's share
            dS = self._lin_bwd(dxn, s["S"], w[a + "output_proj.weight"], grads,
                               a + "output_proj.weight", a + "output_proj.bias")
            gval, gloc, gaw = zeros(M, 256), E(M * 8 * 3 * 4 * 2), E(M * 8 * 3 * 4)
            hip.msda_bwd(s["value"], 256, t["shapes_dev"], t["lsi_dev"], s["loc"], s["aw"], dS, gval,
                         gloc, gaw, B, SN, SN, 3)
            self._acc(dxn, self._lin_bwd(gval, s["x_in"], w[a + "value_proj.weight"], grads,
                                         a + "value_proj.weight", a + "value_proj.bias"))
            d_offaw = E(M, 288)
            hip.msda_offaw_bwd(gloc, gaw, s["aw"], d_offaw, shapes)
            dxq = self._lin_bwd(d_offaw[:, :192], s["xq"], w[a + "sampling_offsets.weight"], grads,
                                a + "sampling_offsets.weight", a + "sampling_offsets.bias")
            self._acc(dxq, self._lin_bwd(d_offaw[:, 192:], s["xq"], w[a + "attention_weights.weight"],
                                         grads, a + "attention_weights.weight",
                                         a + "attention_weights.bias"))
            self._acc(dxn, dxq)
            self._acc(dpos_rows, dxq)
            dx = dxn
            ready(self.group_end["encoder.layers.%d" % i])
        # query_pos = sine table + level_encoding[l] on the tokens of level l
        dle = grads[pd + "level_encoding.weight"]
        for b in range(B):
            for l in range(3):
                r0 = b * SN + start[l]
                hip.colsum(dpos_rows[r0:r0 + N[l]], dle[l], accumulate=True)
        ready(self.group_end["level_encoding"])
        # input convolutions: GroupNorm, then the 1x1 convolution as a linear layer over pixels
        G = head.gn_groups
        stats = E(B * G * 4)
        dX = dx.view(B, SN, 256)
        dfeats = []
        for l in range(3):
            n = N[l]
            conv = t["convs"][l]
            dconv, gx = E(B * n, 256), E(B * n, 256)
            hip.groupnorm_nhwc_bwd(conv, dX[:, start[l]:], w[pd + "input_convs.%d.gn.weight" % l],
                                   dconv, gx, stats, B, n, G, n * 256, SN * 256)
            hip.colsum(gx, grads[pd + "input_convs.%d.gn.weight" % l], accumulate=True)
            C = t["rows"][l][0].shape[1]
            h, wd = shapes[l]
            df = E(B, C, h, wd)
            for b in range(B):
                hip.colsum(dX[b, start[l]:start[l] + n], grads[pd + "input_convs.%d.gn.bias" % l],
                           accumulate=True)
                drows = self._lin_bwd(dconv[b * n:(b + 1) * n], t["rows"][l][b],
                                      w[pd + "input_convs.%d.conv.weight" % l].view(256, C), grads,
                                      pd + "input_convs.%d.conv.weight" % l,
                                      pd + "input_convs.%d.conv.bias" % l)
                hip.transpose(drows, df[b].view(C, n))
            dfeats.append(df)
            ready(self.group_end["input_convs.%d" % l])
        ready(self.flat_numel)
        return dfeats, grads


class BackboneGrad(RelationTailGrad):
    """The backbone's trainable part -- mmdet ResNet-50 / -101 stages 2-4 (`frozen_stages=1`: stem
    and layer1 fixed; BatchNorm frozen, `norm_eval`: configs/mask2former/pairnet.py:9-19) -- taped
    and differentiated from C2 (layer1's output) to C3 / C4 / C5:

        tape = BackboneGrad(det.backbone)
        c3, c4, c5 = tape.forward(c2)                  # channel-last [B, h, w, C] maps
        grads = tape.backward(d_c3, d_c4, d_c5)        # [B, C, h, w], e.g. PixelDecoderGrad's dfeats

    Frozen BatchNorm is folded into the convolutions (W' = W gamma / sqrt(var + eps), as the
    inference path does); the gradient of the un-folded weight is the folded one's times the same
    per-channel factor.  1x1 convolutions are linear layers over pixels (`pn_gemm_f32` on
    transposed operands), the 3x3 convolutions' data gradient is the forward implicit-GEMM kernel on
    the tap-reversed, channel-swapped weight (over the zero-dilated gradient map for the stride-2
    layers), their weight gradient `pn_conv_wgrad_f32` (MFMA outer products over pixels).  The taped
    forward uses the direct implicit-GEMM convolution where inference uses Winograd (same values to
    ~1e-5).  Gradients are named like the backbone's state dict (`layer3.4.conv2.weight`)."""

    def __init__(self, backbone, flat=None, base=0):
        if backbone.device is None or backbone.device.type != "cuda":
            raise RuntimeError("BackboneGrad needs a backbone on an MI355X (.to('cuda:N'))")
        if backbone.w is None:
            backbone._pack()
        self.head, self.dev, self.t = backbone, backbone.device, None
        self._build_layout(backbone, flat, base)
        P = backbone._params
        self.bn_scale, self.bn_scale64 = {}, {}
        for n in self.layout:                      # conv name -> gamma / sqrt(var + eps)
            conv = n[:-len(".weight")]
            bn = conv.replace("conv", "bn") if "downsample" not in conv else conv[:-1] + "1"
            sc = P[bn + ".weight"].double() / torch.sqrt(P[bn + ".running_var"].double() + 1e-5)
            self.bn_scale[n] = sc.float().to(self.dev)
            self.bn_scale64[n] = sc.to(self.dev)

    @staticmethod
    def param_groups(backbone):
        groups = []
        stages = list(enumerate(backbone.stages))
        for i, (planes, blocks) in reversed(stages[1:]):
            for b in reversed(range(blocks)):
                p = "layer%d.%d." % (i + 1, b)
                names = [p + "conv3.weight", p + "conv2.weight", p + "conv1.weight"]
                if b == 0:
                    names.append(p + "downsample.0.weight")
                groups.append((p[:-1], names))
        return groups

    @torch.no_grad()
    @hip.on_device
    def forward(self, c2):
        bb, w, E = self.head, self.head.w, self._E
        B, cin, h, wd = c2.shape
        x = c2.permute(0, 2, 3, 1).contiguous()             # channel-last rows
        blocks_t, outs = [], []
        for i, (planes, blocks) in list(enumerate(bb.stages))[1:]:
            for b in range(blocks):
                p = "layer%d.%d." % (i + 1, b)
                stride = 2 if b == 0 else 1
                hi, wi = h, wd
                if stride == 2:
                    h, wd = (h - 1) // 2 + 1, (wd - 1) // 2 + 1
                s = dict(p=p, x=x, cin=cin, planes=planes, stride=stride, hi=hi, wi=wi, h=h, w=wd)
                s["t1"] = E(B, hi, wi, planes)
                hip.linear(x.view(-1, cin), w[p + "conv1.w"], w[p + "conv1.b"],
                           s["t1"].view(-1, planes), relu=True)
                s["t2"] = E(B, h, wd, planes)
                hip.conv2d_ex(s["t1"], w[p + "conv2.w"], w[p + "conv2.b"], None, s["t2"], B, hi, wi,
                              planes, planes, 3, 3, stride, 1, relu=True)
                if b == 0:
                    idt = E(B, h, wd, planes * 4)
                    hip.conv2d_ex(x, w[p + "downsample.0.w"], w[p + "downsample.0.b"], None, idt, B,
                                  hi, wi, cin, planes * 4, 1, 1, stride, 0)
                else:
                    idt = x
                s["out"] = E(B, h, wd, planes * 4)
                hip.linear(s["t2"].view(-1, planes), w[p + "conv3.w"], w[p + "conv3.b"],
                           s["out"].view(-1, planes * 4), res=idt.view(-1, planes * 4),
                           relu_after=True)
                x, cin = s["out"], planes * 4
                blocks_t.append(s)
            outs.append(x)
        self.t = dict(B=B, blocks=blocks_t)
        return tuple(outs)

    def _conv3x3_bwd(self, s, d_t2, grads, B):
        """d_t2 [B, h, w, planes] (pre-ReLU) -> d t1 (post-ReLU input of the 3x3), + its weight."""
        w, E = self.head.w, self._E
        p, planes, stride = s["p"], s["planes"], s["stride"]
        hi, wi, h, wd = s["hi"], s["wi"], s["h"], s["w"]
        rows_per = max(1, min(h, 8))
        chunks = B * ((h + rows_per - 1) // rows_per)
        part = E(chunks, planes * 9 * planes)
        hip.conv_wgrad(d_t2, s["t1"], part, B, hi, wi, h, wd, planes, planes, 3, stride, 1, rows_per)
        dwp = E(planes * 9 * planes)                          # [co][tap][ci]: the packed layout
        hip.colsum(part, dwp)
        g = grads[p + "conv2.weight"]
        g.copy_(dwp.view(planes, 3, 3, planes).permute(0, 3, 1, 2))      # -> [co][ci][3][3]
        hip.scale_rows(g, self.bn_scale[p + "conv2.weight"])
        wb = E(planes * 9 * planes)
        hip.conv_weight_bwd_layout(w[p + "conv2.w"], wb, planes, 9, planes)
        d_t1 = E(B, hi, wi, planes)
        if stride == 1:
            src = d_t2
        else:
            src = E(B, hi, wi, planes)
            hip.dilate2(d_t2, src, B, hi, wi, h, wd, planes)
        hip.conv2d_ex(src, wb.view(planes, 9 * planes), None, None, d_t1, B, hi, wi, planes, planes,
                      3, 3, 1, 1)
        return d_t1

    @torch.no_grad()
    @hip.on_device
    def backward(self, d_c3, d_c4, d_c5, on_ready=None):
        if self.t is None:
            raise RuntimeError("backward() needs a forward() first")
        w, E, t = self.head.w, self._E, self.t
        B = t["B"]
        ready = on_ready if on_ready is not None else (lambda end: None)
        grads = self._zero_grads()
        nhwc = lambda g: g.to(self.dev, torch.float32).permute(0, 2, 3, 1).contiguous()
        stage_grads = {2: nhwc(d_c3), 3: nhwc(d_c4), 4: nhwc(d_c5)}     # by layer number
        dx = None
        for s in reversed(t["blocks"]):
            p, planes, cin, stride = s["p"], s["planes"], s["cin"], s["stride"]
            hi, wi, h, wd = s["hi"], s["wi"], s["h"], s["w"]
            layer, blk = int(p[5]), int(p[7:-1])
            last_of_stage = blk == self.head.stages[layer - 1][1] - 1
            if last_of_stage:                      # a stage's output also feeds the pixel decoder
                g = stage_grads[layer]
                if dx is not None:
                    self._acc(g, dx)
                dx = g
            dy = E(B, h, wd, planes * 4)
            hip.relu_bwd(dx, s["out"], dy)         # ReLU after the shortcut
            # conv3 (1x1) + BN
            d_t2 = self._lin_bwd(dy.view(-1, planes * 4), s["t2"].view(-1, planes), w[p + "conv3.w"],
                                 grads, p + "conv3.weight", None)
            hip.scale_rows(grads[p + "conv3.weight"], self.bn_scale[p + "conv3.weight"])
            hip.relu_bwd(d_t2, s["t2"].view(-1, planes), d_t2)
            d_t1 = self._conv3x3_bwd(s, d_t2.view(B, h, wd, planes), grads, B)
            hip.relu_bwd(d_t1, s["t1"], d_t1)
            first = layer == 2 and blk == 0        # its input is the frozen layer1's output
            dxin = self._lin_bwd(d_t1.view(-1, planes), s["x"].view(-1, cin), w[p + "conv1.w"], grads,
                                 p + "conv1.weight", None, need_dx=not first)
            hip.scale_rows(grads[p + "conv1.weight"], self.bn_scale[p + "conv1.weight"])
            if blk == 0:                           # projection shortcut (1x1, stride s)
                xs = s["x"]
                if stride == 2:
                    xs = E(B, h, wd, cin)
                    hip.subsample2(s["x"], xs, B, hi, wi, h, wd, cin)
                dxs = self._lin_bwd(dy.view(-1, planes * 4), xs.view(-1, cin),
                                    w[p + "downsample.0.w"], grads, p + "downsample.0.weight", None,
                                    need_dx=not first)
                hip.scale_rows(grads[p + "downsample.0.weight"],
                               self.bn_scale[p + "downsample.0.weight"])
                if not first:
                    if stride == 2:
                        hip.dilate2(dxs, dxin, B, hi, wi, h, wd, cin, accumulate=True)
                    else:
                        self._acc(dxin, dxs)
            else:                                  # identity shortcut
                self._acc(dxin, dy.view(-1, planes * 4))
            dx = None if first else dxin.view(B, hi, wi, cin)
            ready(self.group_end[p[:-1]])
        ready(self.flat_numel)
        return grads
