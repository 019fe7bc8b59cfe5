"""CrossHeadBaseline on MI355X: the reference's sibling head ("PSGFormer+"), sharing the
Mask2Former trunk of `CrossHead2` (head.py) and replacing the Pair Proposal Network by
learned relation queries that cross-attend the pixel memories and are matched to
subject / object queries by cosine argmax.

Mirrors pairnet/models/relation_heads/baseline.py (`CrossHeadBaseline`): constructor
keywords (:25-66), `forward(feats, img_metas)` (:298-443), `get_bboxes` (:967-998),
`_get_bboxes_single` (:1000-1154), `simple_test_bboxes` (:1156-1160) and the state-dict
key names, so `configs/mask2former/baseline_r50_psg.py` and its checkpoint drop in.
Same kernels, same C ABI, no CPU path.

One deliberate default differs from the reference at inference: the reference stacks the
class / mask logits of all nine decoder layers (needed by the training losses) although
`get_bboxes` reads only the last one (:979-986 index [-1]).  `return_all_layers=False`
(default) returns them with a leading dimension of 1, i.e. the last layer only, which
skips eight Q x H/4*W/4 mask GEMMs per image; `return_all_layers=True` reproduces the
reference's (9, B, Q, ...) stacks.
"""
import torch

from . import hip
from .head import CrossHead2


class CrossHeadBaseline(CrossHead2):
    """Drop-in for the reference's `CrossHeadBaseline` (inference half)."""

    RELATION_ORDER = ("self_attn", "norm", "cross_attn", "norm", "ffn", "norm")

    def __init__(self, num_classes, in_channels, num_relations, object_classes=None,
                 predicate_classes=None, num_obj_query=100, num_rel_query=100, use_mask=True,
                 rel_loss_cls=None, sub_id_loss=None, obj_id_loss=None, **kwargs):
        if num_obj_query != num_rel_query:      # baseline.py:92
            raise AssertionError("num_obj_query must equal num_rel_query")
        self.object_classes, self.predicate_classes = object_classes, predicate_classes
        self.return_all_layers = False
        super().__init__(num_classes, in_channels, num_relations, num_obj_query=num_obj_query,
                         num_rel_query=num_rel_query, use_mask=use_mask, **kwargs)

    # ------------------------------------------------------------------ params
    def param_shapes(self):
        """baseline.py:94-194 in registration order."""
        s = super().param_shapes()
        for k in list(s):
            if k.startswith(("rel_query_embed2.", "rel_query_embed3.", "update_importance.")) \
                    or k.endswith(("_query_update.4.weight", "_query_update.4.bias")):
                del s[k]
        s["rel_cls_embed.weight"] = (self.num_relations + 1, 256)
        s["rel_cls_embed.bias"] = (self.num_relations + 1,)
        return s

    def _pack_relation(self, w):
        for i in range(self.num_dec_layers):
            self._pack_vqk(w, "transformer_decoder.layers.%d.attentions.1.attn." % i)
        for i in range(self.num_rel_layers):        # self-attention comes first here
            self._pack_vqk(w, "relation_decoder.layers.%d.attentions.0.attn." % i)

    def _plan_relation(self, pl, E):
        B, Q, R, dev = pl.B, self.num_obj_query, self.num_rel_query, self.device
        BQ, BR, HW2 = B * Q, B * R, pl.HW2
        nd, nc = self.num_dec_layers, self.num_classes + 1
        pl.scr = E(max(hip.attn_scratch_floats(B, max(Q, R), n) for n in pl.N + [Q, R]))
        pl.all_layers = self.return_all_layers
        if pl.all_layers:
            pl.cls_all, pl.MP_all = E(nd, B, Q, nc), E(nd, B, Q, HW2)
            pl.cls, pl.MP = pl.cls_all[nd - 1], pl.MP_all[nd - 1]
        else:
            pl.cls_all, pl.MP_all = pl.cls.view(1, B, Q, nc), pl.MP.view(1, B, Q, HW2)
        nr = self.num_rel_layers
        pl.rKp = [E(B, pl.N[i % 3], 256) for i in range(nr)]
        pl.rVp = [E(B, pl.N[i % 3], 256) for i in range(nr)]
        pl.r, pl.r1, pl.r2, pl.ry = E(BR, 256), E(BR, 256), E(BR, 256), E(BR, 256)
        pl.rQp, pl.ratt, pl.rVQK = E(BR, 256), E(BR, 256), E(BR, 768)
        pl.rh = E(hip.ffn_scratch_floats(BR, self.rel_ffn))
        pl.s1, pl.sn, pl.on, pl.rn = E(BQ, 256), E(BQ, 256), E(BQ, 256), E(BR, 256)
        pl.sub_scores, pl.obj_scores = E(B, R, Q), E(B, R, Q)
        i64 = lambda *s: torch.empty(*s, device=dev, dtype=torch.int64)
        pl.sub_ids, pl.obj_ids = i64(B, R), i64(B, R)
        pl.rel = E(B, R, self.num_relations + 1)
        pl.sub_cls, pl.obj_cls = E(B, R, nc), E(B, R, nc)
        pl.sub_seg, pl.obj_seg = E(B, R, HW2), E(B, R, HW2)

    # ------------------------------------------------------------------ stages
    def pair_positions(self, pl=None):
        """(sub_ids, obj_ids): the object-query rows matched to each relation query
        (baseline.py:389-401)."""
        pl = pl if pl is not None else self._last_plan
        return pl.sub_ids, pl.obj_ids

    def _kv_problems(self, pl):
        """Stage A additionally projects the memories for the relation decoder's six
        cross-attentions (baseline.py:372-384: key = value = memory + level embedding,
        key_pos = sine encoding, levels cycling like the object decoder)."""
        probs = super()._kv_problems(pl)
        for i in range(self.num_rel_layers):
            probs += self._memory_kv(pl, "relation_decoder.layers.%d.attentions.1.attn." % i,
                                     i % 3, pl.rKp[i], pl.rVp[i])
        return probs

    def _stage_b(self, pl):
        self._object_decoder(pl, all_layers=pl.all_layers)
        self._relation_stage(pl)

    def _relation_stage(self, pl):
        w, B, Q, R = self.w, pl.B, self.num_obj_query, self.num_rel_query
        # ---- relation decoder over the pixel memories (baseline.py:363-386) ----
        pl.r.view(B, R, 256).copy_(w["rel_query_feat.weight"].unsqueeze(0).expand(B, R, 256))
        rpos = w["rel_query_embed.weight"]
        for i in range(self.num_rel_layers):
            l = i % 3
            self._layer("relation_decoder.layers.%d." % i, pl.r, rpos, pl.r1, pl.r2, pl.ry,
                        pl.rQp, pl.rVQK, pl.ratt, pl.rh, pl.rKp[i], 256, pl.rVp[i], 256, pl.N[l],
                        B, R, None, None, pl.scr, self.rel_ffn, self_first=True)
        # ---- query matching (:388-399) ----
        for mlp, dst in (("sub_query_update", pl.sn), ("obj_query_update", pl.on)):
            hip.linear(pl.q, w[mlp + ".0.weight"], w[mlp + ".0.bias"], pl.s1, relu=True)
            hip.linear(pl.s1, w[mlp + ".2.weight"], w[mlp + ".2.bias"], dst)
            hip.l2normalize(dst, dst)
        hip.l2normalize(pl.r, pl.rn)
        for emb, sc, ids in ((pl.sn, pl.sub_scores, pl.sub_ids), (pl.on, pl.obj_scores, pl.obj_ids)):
            hip.gemm(pl.rn, emb, sc, M=R, N=Q, K=256, lda=256, ldw=256, ldc=Q, batch=B,
                     sA=R * 256, sW=Q * 256, sC=R * Q)
            hip.row_argmax(sc, ids, B * R, Q)
        hip.linear(pl.r, w["rel_cls_embed.weight"], w["rel_cls_embed.bias"], pl.rel.view(B * R, -1))
        # ---- output gathers (:401-430) ----
        nc = self.num_classes + 1
        hip.gather_rows(pl.cls, pl.sub_ids, pl.sub_cls, B, Q, R, nc)
        hip.gather_rows(pl.cls, pl.obj_ids, pl.obj_cls, B, Q, R, nc)
        hip.gather_rows(pl.MP, pl.sub_ids, pl.sub_seg, B, Q, R, pl.HW2)
        hip.gather_rows(pl.MP, pl.obj_ids, pl.obj_seg, B, Q, R, pl.HW2)

    def _outputs(self, pl):
        B, Q, R = pl.B, self.num_obj_query, self.num_rel_query
        H2, W2 = pl.hw2
        n = pl.MP_all.shape[0]
        return (dict(sub=pl.sub_cls, obj=pl.obj_cls, cls=pl.cls_all, rel=pl.rel,
                     subject_scores=pl.sub_scores, object_scores=pl.obj_scores),
                dict(mask=pl.MP_all.view(n, B, Q, H2, W2), sub_seg=pl.sub_seg.view(B, R, H2, W2),
                     obj_seg=pl.obj_seg.view(B, R, H2, W2)))

    # ------------------------------------------------------- post-processing
    @torch.no_grad()
    @hip.on_device
    def get_bboxes(self, cls_scores, mask_preds, img_metas, rescale=False):
        """baseline.py:967-998."""
        self._pan_jobs = []
        res = CrossHead2.ResultList(self._get_bboxes_single(
            mask_preds["mask"][-1, i], cls_scores["cls"][-1, i], cls_scores["sub"][i],
            cls_scores["obj"][i], cls_scores["rel"][i], mask_preds["sub_seg"][i],
            mask_preds["obj_seg"][i], img_metas[i]["img_shape"],
            img_metas[i]["scale_factor"], rescale) for i in range(len(img_metas)))
        res.panoptic_jobs = tuple(self._pan_jobs)
        return res

    def _get_bboxes_single(self, all_masks, all_cls, s_cls, o_cls, r_cls, s_seg, o_seg,
                           img_shape, scale_factor, rescale=False):
        """baseline.py:1000-1154 on the device, asynchronously."""
        assert len(s_cls) == len(o_cls) == len(r_cls)
        dev = all_cls.device
        R, Q, nc = len(r_cls), all_cls.shape[0], all_cls.shape[-1]
        nrel1 = r_cls.shape[-1]                       # num_relations + 1
        k = self.test_cfg.get("max_per_img", self.num_obj_query)
        H0 = round(img_shape[0] / scale_factor[1])
        W0 = round(img_shape[1] / scale_factor[0])
        h, wd = all_masks.shape[-2:]
        i64 = lambda *s: torch.empty(*s, device=dev, dtype=torch.int64)
        f32 = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        # subject / object labels of every relation query (:1026-1031)
        s_lab, o_lab, sc_tmp = i64(R), i64(R), f32(R)
        hip.cls_argmax(s_cls.contiguous(), s_lab, sc_tmp, R, nc)
        hip.cls_argmax(o_cls.contiguous(), o_lab, sc_tmp, R, nc)
        # rank the R x num_relations (query, predicate) pairs (:1033-1037)
        probs, fg = f32(R, nrel1), f32(R * (nrel1 - 1))
        hip.softmax_fg(r_cls.contiguous(), probs, fg, R, nrel1)
        r_idx, tri, rem = i64(k), i64(k), i64(k)
        hip.topk(fg, r_idx, tri, rem, 1, R * (nrel1 - 1), nrel1 - 1, k)
        labels, r_labels, r_scores, r_dists = i64(2 * k), i64(k), f32(k), f32(k, nrel1)
        hip.triplet_finish(s_lab, o_lab, probs, tri, rem, labels, r_labels, r_scores, r_dists, k,
                           nrel1)
        # masks of the ranked triplets at the original image size (:1048-1073)
        seg = f32(2 * k, h * wd)
        hip.gather_rows(s_seg.contiguous(), tri, seg[:k], 1, R, k, h * wd)
        hip.gather_rows(o_seg.contiguous(), tri, seg[k:], 1, R, k, h * wd)
        masks_u8 = torch.empty(2 * k, H0, W0, device=dev, dtype=torch.uint8)
        hip.bilinear_planar_gt0(seg, masks_u8, 2 * k, h, wd, H0, W0)
        masks = masks_u8.view(torch.bool)
        # panoptic map (:1050-1131): the same code as CrossHead2's
        all_labels, all_scores = i64(Q), f32(Q)
        hip.cls_argmax(all_cls.contiguous(), all_labels, all_scores, Q, nc)
        state = torch.empty(hip.panoptic_state_bytes(), device=dev, dtype=torch.uint8)
        up = f32(Q, H0 * W0)
        area = torch.empty(256, device=dev, dtype=torch.int32)
        pan = i64(H0 * W0)
        hip.panoptic_device(all_masks.contiguous(), all_labels, all_scores, Q, nc - 1, h, wd, H0,
                            W0, state, up, area, pan)
        self._pan_jobs.append((state, up, area, pan, H0, W0))
        # the reference fills det_bboxes with torch.rand as "dummy bboxes for eval" (:1134)
        det_bboxes = torch.zeros((2 * k, 5), device=dev)
        rel_pairs = torch.arange(2 * k, dtype=torch.int).reshape(2, -1).T
        return (det_bboxes, labels, rel_pairs, masks, pan.view(H0, W0), r_scores, r_labels,
                r_dists)
