"""Generate tests/golden/*.npz from the REFERENCE's own code (build container only).

    python -m oracle.make_golden [--only NAME]

TEST INFRASTRUCTURE.  The reference head (pairnet_head.py) and Matrix Learner
(cnn_factory.py) are imported from /root/reference under the name-only shims of
oracle/ref_shim.py and run on seeded weights / inputs (oracle/seeded.py).  Only
arrays are written: inputs (or their seed + checksum) and the reference's outputs.
No reference source text is stored.

Fixtures
  convtiny   G2  reference ConvTiny on a seeded (2,100,100) importance matrix
  ppn        G1  reference sub/obj MLPs + ConvTiny + top-k on a stored query tensor
  reldec     G3  pair features -> relation logits through the reference head's
                 relation decoder loop (pairnet_head.py:353-378)
  fwdhead    G4  reference forward_head (pairnet_head.py:216-258) on a reduced mask
                 feature: class / mask logits, boolean attention mask, pre-threshold
                 resized logits (so that bits next to the threshold can be exempted)
  declayer   G5  one masked-attention decoder layer of the reference head
                 (transformer_decoder.layers[0], attn_masks=[mask, None]) at reduced K
  msda       G6  deformable sampling on a reduced pyramid (restated CPU formula;
                 unpinned against mmcv)
  e2e_small  G7  whole head + get_bboxes on a 96x128 image, batch 2
  e2e_full   G8  whole head on the 800x1333 north-star shape, batch 1 (statistics,
                 logits and indices only)
  baseline_small  the sibling head CrossHeadBaseline (relation_heads/baseline.py):
                 forward + get_bboxes on a 96x128 image, batch 2
  psgtr2_small   the sibling head PSGTrHead2 (relation_heads/psgtr_head2.py): forward +
                 get_bboxes on a 96x128 image, batch 1 (the reference's get_bboxes only
                 handles one image per call)
  bbox_small / bbox_full   the sibling head CrossHeadBBox (relation_heads/pairnet_bbox_head.py,
                 config configs/deformable_detr/cross_r101_vg.py) over the restated neck and
                 Deformable-DETR trunk: forward + get_bboxes at 160x192 (batch 2) and at
                 800x1333 (batch 1), on inputs chosen so that the three index selections of
                 the path (300 proposals, 100 kept queries, 100 pairs) are separated from
                 fp32 rounding noise
"""
import argparse
import os
import sys
import time
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

sys.dont_write_bytecode = True
from . import layers as L  # noqa: E402
from . import ref_shim, seeded  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "tests", "golden")
WEIGHT_SEED = 20240917


def _np(t):
    return t.detach().cpu().numpy()


def build_ref(seed=WEIGHT_SEED):
    cfg = ref_shim.reference_head_cfg()
    head = ref_shim.build_reference_head(cfg)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in head.state_dict().items())
    sd = seeded.seeded_state_dict(shapes, seed)
    head.load_state_dict(sd, strict=True)
    return head, sd


def topk_gaps(importance, k):
    """min gap between consecutive sorted scores among the top k+1 (order AND
    membership of the top-k are stable under perturbations well below this)."""
    v = importance.flatten(-2, -1).sort(dim=-1, descending=True)[0][..., :k + 1]
    return (v[..., :-1] - v[..., 1:]).min(-1)[0]


def gen_convtiny():
    net = ref_shim.reference_conv_tiny()
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    sd = seeded.seeded_state_dict(shapes, 11)
    net.load_state_dict(sd)
    x = seeded.uniform(np.random.default_rng(12), (2, 100, 100), -1.0, 1.0)
    with torch.no_grad():
        y = net(x)
    np.savez_compressed(os.path.join(OUT, "convtiny.npz"), weight_seed=11,
                        weight_crc=seeded.checksum(sd), x=_np(x), y=_np(y))


def gen_ppn(head, sd):
    rng = np.random.default_rng(21)
    q = seeded.uniform(rng, (100, 1, 256), -2.0, 2.0)   # last-layer query_feat (Q,B,C)
    with torch.no_grad():  # pairnet_head.py:322-340 with the reference's modules
        s = F.normalize(head.sub_query_update(q).transpose(0, 1), p=2, dim=-1, eps=1e-12)
        o = F.normalize(head.obj_query_update(q).transpose(0, 1), p=2, dim=-1, eps=1e-12)
        raw = torch.matmul(s, o.transpose(1, 2))
        imp = head.update_importance(raw)
        _, idx = torch.topk(imp.flatten(-2, -1), k=head.num_rel_query)
        sub = torch.div(idx, head.num_obj_query, rounding_mode="trunc")
        obj = torch.remainder(idx, head.num_obj_query)
    np.savez_compressed(os.path.join(OUT, "ppn.npz"), weight_seed=WEIGHT_SEED,
                        weight_crc=seeded.checksum(sd), query_feat=_np(q),
                        importance_raw=_np(raw), importance=_np(imp), topk_idx=_np(idx),
                        sub_pos=_np(sub), obj_pos=_np(obj),
                        min_gap=_np(topk_gaps(imp, head.num_rel_query)))


def gen_reldec(head, sd):
    rng = np.random.default_rng(31)
    bs = 2
    pair = seeded.uniform(rng, (200, bs, 256), -2.0, 2.0)
    with torch.no_grad():  # pairnet_head.py:353-378
        r = head.rel_query_feat.weight.unsqueeze(1).repeat((1, bs, 1))
        e1 = head.rel_query_embed.weight.unsqueeze(1).repeat((1, bs, 1))
        e2 = head.rel_query_embed2.weight.unsqueeze(1).repeat((1, bs, 1))
        e3 = head.rel_query_embed3.weight.unsqueeze(1).repeat((1, bs, 1))
        for layer in head.relation_decoder.layers:
            r = layer(query=r, key=pair, value=pair, query_pos=e1, key_pos=e2,
                      value_pos=e3, query_key_padding_mask=None, key_padding_mask=None)
        rel = head.rel_cls_embed(r.transpose(0, 1))
    np.savez_compressed(os.path.join(OUT, "reldec.npz"), weight_seed=WEIGHT_SEED,
                        weight_crc=seeded.checksum(sd), pair_feat=_np(pair), rel_preds=_np(rel))


def gen_fwdhead(head, sd):
    rng = np.random.default_rng(51)
    dec = seeded.uniform(rng, (100, 2, 256), -2.0, 2.0)
    mf = seeded.uniform(rng, (2, 256, 16, 24), -1.0, 1.0)
    with torch.no_grad():
        cls, mask, attn = head.forward_head(dec, mf, (8, 12))
        resized = F.interpolate(mask, (8, 12), mode="bilinear", align_corners=False)
    np.savez_compressed(os.path.join(OUT, "fwdhead.npz"), weight_seed=WEIGHT_SEED,
                        weight_crc=seeded.checksum(sd), decoder_out=_np(dec), mask_feature=_np(mf),
                        cls_pred=_np(cls), mask_pred=_np(mask),
                        attn_mask=np.packbits(_np(attn)), attn_shape=np.array(attn.shape),
                        resized_logits=_np(resized))


def gen_declayer(head, sd):
    rng = np.random.default_rng(61)
    Q, K, bs = 100, 96, 2
    q = seeded.uniform(rng, (Q, bs, 256), -1.0, 1.0)
    mem = seeded.uniform(rng, (K, bs, 256), -1.0, 1.0)
    # positional terms are batch-independent in the model (embedding / sine table repeated)
    qpos = seeded.uniform(rng, (Q, 1, 256), -1.0, 1.0).repeat(1, bs, 1)
    kpos = seeded.uniform(rng, (K, 1, 256), -1.0, 1.0).repeat(1, bs, 1)
    mask = torch.from_numpy(rng.random((bs, Q, K)) < 0.5)
    mask[:, 7, :] = False                      # the all-masked-row fix leaves such rows open
    attn = mask.unsqueeze(1).repeat(1, head.n_heads, 1, 1).flatten(0, 1)
    with torch.no_grad():
        out = head.transformer_decoder.layers[0](
            query=q, key=mem, value=mem, query_pos=qpos, key_pos=kpos, attn_masks=[attn, None],
            query_key_padding_mask=None, key_padding_mask=None)
    np.savez_compressed(os.path.join(OUT, "declayer.npz"), weight_seed=WEIGHT_SEED,
                        weight_crc=seeded.checksum(sd), query=_np(q), query_pos=_np(qpos),
                        memory=_np(mem), key_pos=_np(kpos), mask=np.packbits(_np(mask)),
                        mask_shape=np.array(mask.shape), out=_np(out))


def gen_msda():
    rng = np.random.default_rng(41)
    shapes = [(3, 4), (6, 8), (12, 16)]
    n = sum(h * w for h, w in shapes)
    bs = 2
    value = seeded.uniform(rng, (bs, n, 8, 32), -1.0, 1.0)
    off = seeded.uniform(rng, (bs, n, 8, 3, 4, 2), -3.0, 3.0)
    logits = seeded.uniform(rng, (bs, n, 8, 12), -2.0, 2.0)
    refs = []
    for h, w in shapes:
        yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5,
                                torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
        refs.append(torch.stack([xx.reshape(-1) / w, yy.reshape(-1) / h], -1))
    ref = torch.cat(refs, 0)[None, :, None].repeat(bs, 1, 3, 1)
    norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)
    loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    aw = logits.softmax(-1).view(bs, n, 8, 3, 4)
    out = L.msda_core(value, shapes, loc, aw)
    np.savez_compressed(os.path.join(OUT, "msda.npz"), shapes=np.array(shapes),
                        value=_np(value), offsets=_np(off), logits=_np(logits), out=_np(out))


MF_BIAS = "pixel_decoder.mask_feature.bias"


def calibrate_mask_bias(head, feats):
    """Shift mask_feature.bias so that every mask-feature channel is zero-mean over the
    pixels of THIS input: mask logits then change sign over the image and the boolean
    attention masks come out ~50 % dense (seeded weights alone give ~2 %, which would
    leave the masked-attention path untested).  The 256 floats are stored in the
    fixture as an override on top of the seeded weights."""
    with torch.no_grad():
        mf, _ = head.pixel_decoder(feats)
        head.pixel_decoder.mask_feature.bias -= mf.mean(dim=(0, 2, 3))
    return head.pixel_decoder.mask_feature.bias.detach().clone()


def _run_e2e(head, feats, metas):
    with torch.no_grad():
        t = time.time()
        cls, masks = head.forward(feats, metas)
        dt = time.time() - t
        _, idx = torch.topk(cls["importance"].flatten(-2, -1), k=head.num_rel_query)
    return cls, masks, idx, dt


def gen_e2e_small(head, sd):
    H, W, bs = 96, 128, 2
    feats = seeded.seeded_feats(51, bs, H, W)
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0, 2.0, 2.0, 2.0])] * bs
    mf_bias = calibrate_mask_bias(head, feats)
    cls, masks, idx, _ = _run_e2e(head, feats, metas)
    with torch.no_grad():
        res = head.get_bboxes(cls, masks, metas)
    out = dict(weight_seed=WEIGHT_SEED, weight_crc=seeded.checksum(sd), feat_seed=51,
               feat_crc=seeded.checksum(feats), height=H, width=W, batch=bs,
               topk_idx=_np(idx), min_gap=_np(topk_gaps(cls["importance"], 100)),
               mask_neg_frac=float((masks["mask"] < 0).float().mean()))
    out["override_" + MF_BIAS] = _np(mf_bias)
    for k, v in cls.items():
        out["cls_" + k] = _np(v)
    for k, v in masks.items():
        out["mask_" + k] = _np(v)
    for i, r in enumerate(res):
        for name, v in zip(("bboxes", "labels", "rel_pairs", "masks", "pan_img", "r_scores",
                            "r_labels", "r_dists"), r):
            out["res%d_%s" % (i, name)] = np.packbits(_np(v)) if name == "masks" else _np(v)
        out["res%d_masks_shape" % i] = np.array(r[3].shape)
    np.savez_compressed(os.path.join(OUT, "e2e_small.npz"), **out)


def gen_e2e_full(head, sd):
    H, W = 800, 1333
    feats = seeded.seeded_feats(61, 1, H, W)
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.083, 2.083, 2.083, 2.083])]
    mf_bias = calibrate_mask_bias(head, feats)
    cls, masks, idx, dt = _run_e2e(head, feats, metas)
    m = masks["mask"]
    probe = torch.from_numpy(np.random.default_rng(62).integers(0, m.numel(), 4096))
    np.savez_compressed(
        os.path.join(OUT, "e2e_full.npz"), weight_seed=WEIGHT_SEED,
        weight_crc=seeded.checksum(sd), feat_seed=61, feat_crc=seeded.checksum(feats),
        height=H, width=W, rel=_np(cls["rel"]), cls=_np(cls["cls"]),
        importance=_np(cls["importance"]), topk_idx=_np(idx),
        min_gap=_np(topk_gaps(cls["importance"], 100)),
        mask_probe_idx=_np(probe), mask_probe=_np(m.flatten()[probe]),
        mask_mean=float(m.mean()), mask_absmean=float(m.abs().mean()),
        mask_neg_frac=float((m < 0).float().mean()), ref_seconds=dt,
        feat_shapes=np.array([f.shape for f in feats]), **{"override_" + MF_BIAS: _np(mf_bias)})
    print("e2e_full: reference forward %.1f s on %d threads" % (dt, torch.get_num_threads()))


# --------------------------------------------------------------------------- separated
# "Separated" fixtures (e2e_small_sep, e2e_full_sep, ppn_sep): the same seeded weights
# plus a few stored edits (oracle/seeded.py "ops") chosen so that the reference's top-k
# pair list is STABLE -- every gap between consecutive scores among the top k+1 is far
# above what fp32 re-association can move a score by -- and strict index equality is a
# meaningful assertion.  With purely random weights it is not, for two measured reasons
# (LABNOTES.md section 3): (1) the nine post-norm decoder layers collapse all queries onto
# one common vector (diversity 1-3 % of the norm), so the pair scores differ by 1e-5 while
# a single flipped attention-mask bit (a mask logit within 1e-6 of zero; a few per 800x1333
# forward, also between the reference's own fp32 and fp64 runs) moves them by 1e-6; (2) a
# random 3-layer 7x7x64 CNN ends in 3136-term sums whose rounding error is 3e-5 of its
# output spread, about the typical smallest gap of 100 order statistics.  The edits give
# the head the properties of a trained one: cross-attention that does not add a common
# vector to every query (value bias centred on this input, output projection scaled so
# that queries stay dominated by their own learned embeddings), subject / object
# embeddings without the common component and of low dimension (wide cosine spread), and
# a Matrix Learner with an identity path beside its (fully random) weights.  A cheap
# search over the seed of the two last MLP layers then picks the widest minimum gap.
XATTN_GAIN = float(os.environ.get("SEP_XATTN", 0.1))
MIN_MARGIN = 10.0   # min gap / fp32-vs-fp64 score error the generator insists on
EMB_DIM, SEARCH = int(os.environ.get("SEP_DIM", 16)), int(os.environ.get("SEP_SEARCH", 1000))


def _ppn_scores(sd, q, dtype):
    """pairnet_head.py:322-333 from a state dict (used by the seed search)."""
    W = lambda n: sd[n].to(dtype)
    q = q.to(dtype)

    def mlp(p):
        x = F.relu(F.linear(q, W(p + ".0.weight"), W(p + ".0.bias")))
        x = F.relu(F.linear(x, W(p + ".2.weight"), W(p + ".2.bias")))
        x = F.linear(x, W(p + ".4.weight"), W(p + ".4.bias"))
        return F.normalize(x.transpose(0, 1), p=2, dim=-1, eps=1e-12)
    x = torch.matmul(mlp("sub_query_update"), mlp("obj_query_update").transpose(1, 2))[:, None]
    for i in range(3):
        p = "update_importance.conv_layers.%d.0." % i
        x = F.conv2d(x, W(p + "weight"), W(p + "bias"), padding=3)
        if i < 2:
            x = F.relu(x)
    return x[:, 0]


def _top_noise(imp32, imp64, k):
    """Largest |fp32 - fp64| score difference among the 2k highest-scoring pairs of each
    image (the pairs whose order the top-k list depends on)."""
    a, b = imp32.flatten(-2, -1).double(), imp64.flatten(-2, -1).double()
    top = a.topk(2 * k, dim=-1)[1]
    return float((a - b).abs().gather(-1, top).max())


def separate(oracle_cls, cfg, sd0, feats, metas, q_override=None, log=print, feats64=None,
             trunk_only=False):
    """Returns (ops, report) for the seeded state dict `sd0` on this input.  With
    `q_override` (Q,B,256) only the PPN edits are made (the ppn_sep fixture).  `feats64`:
    the features of an fp64 evaluation of whatever produced `feats` (a backbone), so that
    the noise estimate of the seed search includes the producer's rounding.  `trunk_only`:
    stop after the decoder edits (everything the attention masks depend on; the seed screen
    of gen_e2e_sep uses it)."""
    import copy
    k = cfg["num_rel_query"]
    ops = {}

    def build(dtype=torch.float32):
        h = oracle_cls(**cfg).eval()
        h.load_state_dict(seeded.apply_ops(dict(sd0), ops), strict=True)
        return h.to(dtype)

    def queries():
        out = []
        for dtype in (torch.float32, torch.float64):
            trace = {}
            fin = feats64 if (dtype == torch.float64 and feats64 is not None) else feats
            build(dtype).forward([f.to(dtype) for f in fin], metas, trace=trace)
            out.append(trace["query_feat"])
        return out

    with torch.no_grad():
        if q_override is None:
            h = build()
            ops["override_" + MF_BIAS] = _np(calibrate_mask_bias(h, feats))
            _, mems = h.pixel_decoder(feats)
            hooks = []
            for i, layer in enumerate(h.transformer_decoder.layers):
                # cross-attention: values centred over this input's memory, output scaled
                a = layer.attentions[0].attn
                m = mems[i % 3].flatten(2).mean(dim=(0, 2)) + h.level_embed.weight[i % 3]
                a.in_proj_bias[512:] = -(a.in_proj_weight[512:] @ m)
                a.out_proj.weight *= XATTN_GAIN
                ops["scale_transformer_decoder.layers.%d.attentions.0.attn.out_proj.weight" % i] \
                    = np.array([XATTN_GAIN, 0, 256])
                # self-attention values and the FFN output centred over the queries, on the
                # fly during one calibration pass (each layer sees the calibrated earlier ones)
                def centre_v(mod, args):
                    mod.attn.in_proj_bias[512:] = -(mod.attn.in_proj_weight[512:]
                                                    @ args[0].mean(dim=(0, 1)))
                def centre_ffn(mod, args):
                    mod.layers[1].bias -= mod.layers(args[0]).mean(dim=(0, 1))
                hooks.append(layer.attentions[1].register_forward_pre_hook(centre_v))
                hooks.append(layer.ffns[0].register_forward_pre_hook(centre_ffn))
            h.forward(feats, metas)
            for hk in hooks:
                hk.remove()
            for i, layer in enumerate(h.transformer_decoder.layers):
                p = "override_transformer_decoder.layers.%d." % i
                for j in (0, 1):
                    ops[p + "attentions.%d.attn.in_proj_bias" % j] = \
                        _np(layer.attentions[j].attn.in_proj_bias)
                ops[p + "ffns.0.layers.1.bias"] = _np(layer.ffns[0].layers[1].bias)
            if trunk_only:
                return ops, {}
            q32, q64 = queries()
        else:
            # (a pair: the fp32 and the fp64 run's queries, so that the noise estimate
            # includes what the trunk's rounding does to them)
            q32, q64 = q_override if isinstance(q_override, tuple) \
                else (q_override, q_override.double())
        log("queries: common component %.2f, diversity %.2f, fp32-vs-fp64 error %.2e"
            % (q64.mean(0).norm(dim=-1).mean(), (q64 - q64.mean(0, keepdim=True)).norm(dim=-1).mean(),
               (q32.double() - q64).abs().max()))
        sd = seeded.apply_ops(dict(sd0), ops)
        for mlp in ("sub_query_update", "obj_query_update"):
            pre = F.linear(q32, sd[mlp + ".0.weight"], sd[mlp + ".0.bias"])
            ops["override_%s.0.bias" % mlp] = _np(sd[mlp + ".0.bias"] - pre.mean(dim=(0, 1)))
            ops["keeprows_%s.4.weight" % mlp] = np.array(EMB_DIM)
            ops["keeprows_%s.4.bias" % mlp] = np.array(EMB_DIM)
        ops["mlearner_skip"] = np.array(1.0)
        best = None
        for s in range(SEARCH):
            ops["reseed_sub_query_update.4.weight"] = np.array(7000 + 2 * s)
            ops["reseed_obj_query_update.4.weight"] = np.array(7001 + 2 * s)
            sd = seeded.apply_ops(dict(sd0), ops)
            i32 = _ppn_scores(sd, q32, torch.float32)
            gap = float(topk_gaps(i32, k).min())
            if best is not None and gap < 0.5 * best[1]:
                continue   # (the fp64 pass is the expensive half)
            noise = _top_noise(i32, _ppn_scores(sd, q64, torch.float64), k)
            if best is None or gap / noise > best[1] / best[2]:
                best = (s, gap, noise)
        s, gap, noise = best
        ops["reseed_sub_query_update.4.weight"] = np.array(7000 + 2 * s)
        ops["reseed_obj_query_update.4.weight"] = np.array(7001 + 2 * s)
        # exact power-of-two output gain: min gap >= 2e-4 without any new rounding
        gain = 2.0 ** max(0, int(np.ceil(np.log2(2e-4 / gap))))
        for n in ("weight", "bias"):
            ops["scale_update_importance.conv_layers.2.0." + n] = np.array([gain, 0, 1])
        log("PPN seed %d of %d: min gap %.3e x %g, fp32-vs-fp64 score error %.3e -> margin %.0f"
            % (s, SEARCH, gap, gain, noise, gap / noise))
    return ops, dict(min_gap=gap * gain, noise=noise * gain)


def _sep_check(head_o64, head, feats, metas, cls, idx, k):
    """The recorded run against the fp64 evaluation of the same head: how far fp32
    arithmetic alone moves the scores, and that the top-k list survives it."""
    t64 = {}
    c64, _ = head_o64.forward([f.double() for f in feats], metas, trace=t64)
    noise = _top_noise(cls["importance"], c64["importance"], k)
    gap = float(topk_gaps(cls["importance"], k).min())
    assert torch.equal(t64["topk_idx"], idx), "top-k list differs between fp32 and fp64"
    assert gap >= 1e-4 and gap >= MIN_MARGIN * noise, (gap, noise)
    return gap, noise


def _min_resized_logit(head, feats, metas):
    """Smallest |resized mask logit| over all layers of one forward (see stable_feat_seed)."""
    real, seen = F.interpolate, []

    def watch(x, *a, **k):
        y = real(x, *a, **k)
        if k.get("mode", None) == "bilinear" and y.shape[-2:] != tuple(feats[0].shape[-2:]):
            seen.append(float(y.abs().min()))
        return y
    F.interpolate = watch
    try:
        with torch.no_grad():
            head.forward(feats, metas)
    finally:
        F.interpolate = real
    return min(seen)


def gen_e2e_sep(name, H, W, bs, feat_seed, sf, screen=0.0):
    """`screen` > 0 (the 96x128 fixture): take the first feature seed >= feat_seed whose run
    keeps every resized mask logit at least that far from zero (6e-5: the ~5e5 logits of a run
    put their minimum at ~1e-5 typically, one seed in ~20 passes).  With 12 / 48 / 192 keys per
    level ONE flipped attention-mask bit moves the importance scores by ~2e-3, and a logit
    within 1e-5 of zero flips under any fp32 re-association (round 3: the exp2-based softmax
    flipped one at 7e-6 on the unscreened seed 53).  At 800x1333 (6.6 M logits per layer) no
    seed can be screened and none needs to be: one bit among 16 700 keys moves a query by
    1e-5."""
    from .head import OracleCrossHead2
    cfg = ref_shim.reference_head_cfg()
    cfg.pop("type", None)
    head = ref_shim.build_reference_head()
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in head.state_dict().items())
    sd0 = seeded.seeded_state_dict(shapes, WEIGHT_SEED)
    metas = [dict(img_shape=(H, W, 3), scale_factor=[sf] * 4)] * bs
    while screen > 0:      # (the PPN edits do not reach the mask logits: screen on the trunk)
        feats = seeded.seeded_feats(feat_seed, bs, H, W)
        ops, _ = separate(OracleCrossHead2, dict(cfg), sd0, feats, metas, trunk_only=True)
        head.load_state_dict(seeded.apply_ops(dict(sd0), ops), strict=True)
        m = _min_resized_logit(head, feats, metas)
        print("%s: feat seed %d: smallest |resized mask logit| %.2e" % (name, feat_seed, m),
              flush=True)
        if m >= screen:
            break
        feat_seed += 1
    feats = seeded.seeded_feats(feat_seed, bs, H, W)
    ops, rep = separate(OracleCrossHead2, dict(cfg), sd0, feats, metas)
    sd = seeded.apply_ops(dict(sd0), ops)
    head.load_state_dict(sd, strict=True)
    cls, masks, idx, dt = _run_e2e(head, feats, metas)
    o64 = OracleCrossHead2(**cfg).eval()
    o64.load_state_dict(sd)
    gap, noise = _sep_check(o64.double(), head, feats, metas, cls, idx, head.num_rel_query)
    print("%s: min gap %.3e, fp32-vs-fp64 importance error %.3e (margin %.0f)"
          % (name, gap, noise, gap / noise))
    with torch.no_grad():
        res = head.get_bboxes(cls, masks, metas)
    m = masks["mask"]
    probe = torch.from_numpy(np.random.default_rng(feat_seed + 1).integers(0, m.numel(), 4096))
    sub = torch.div(idx, head.num_obj_query, rounding_mode="trunc")
    out = dict(weight_seed=WEIGHT_SEED, weight_crc=seeded.checksum(sd), feat_seed=feat_seed,
               feat_crc=seeded.checksum(feats), height=H, width=W, batch=bs, img_scale=sf,
               rel=_np(cls["rel"]), cls=_np(cls["cls"]), importance=_np(cls["importance"]),
               sub=_np(cls["sub"]), obj=_np(cls["obj"]),
               topk_idx=_np(idx), sub_pos=_np(sub), obj_pos=_np(idx - sub * head.num_obj_query),
               min_gap=gap, fp64_noise=noise,
               mask_probe_idx=_np(probe), mask_probe=_np(m.flatten()[probe]),
               mask_neg_frac=float((m < 0).float().mean()), ref_seconds=dt, **ops)
    for i, r in enumerate(res):
        out["res%d_labels" % i] = _np(r[1])
        out["res%d_r_dists" % i] = _np(r[7])
        out["res%d_pan_img" % i] = _np(r[4]).astype(np.int32)
        rows = np.arange(0, r[3].shape[0], 1 if H * W < 100000 else 10)   # (fixture size)
        out["res%d_masks_rows" % i] = rows
        out["res%d_masks" % i] = np.packbits(_np(r[3])[rows])
        out["res%d_masks_shape" % i] = np.array((len(rows),) + tuple(r[3].shape[1:]))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


# ---- image -> triplets fixtures (round 3): the backbone in front of the separated head ----
# BASELINE.json configs[1] / configs[3] name a BACKBONE + head, and PSGTr.simple_test
# (psgtr.py:148-156) starts from the image tensor.  These fixtures record the reference head
# on the features of the oracle backbone (oracle/backbone.py / oracle/swin.py, both pinned to
# HuggingFace transformers' implementations) for a seeded image, separated like the
# `*_sep` fixtures above, plus probes of the backbone features themselves.
def _backbone_oracle(kind, seed):
    if kind == "r50":
        from .backbone import OracleResNet50, seeded_backbone_state
        net = OracleResNet50()
        sd = seeded_backbone_state(seed)
        net.load_state_dict(sd)
        return net, sd, (256, 512, 1024, 2048)
    from .swin import OracleSwin, seeded_swin_state
    dims = dict(swinL=dict(embed_dims=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48),
                           window_size=12),
                swinB=dict(embed_dims=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32),
                           window_size=12))[kind]
    net = OracleSwin(**dims)
    sd = seeded_swin_state(net, seed)
    net.load_state_dict(sd)
    e = dims["embed_dims"]
    return net, sd, (e, 2 * e, 4 * e, 8 * e)


def gen_e2e_image(name, kind, H, W, bs, img_seed, bb_seed, sf, num_obj_query=100):
    import copy
    from .head import OracleCrossHead2
    backbone, bsd, chans = _backbone_oracle(kind, bb_seed)
    img = seeded.uniform(np.random.default_rng(img_seed), (bs, 3, H, W), -2.0, 2.0)
    t0 = time.time()
    feats = [f.contiguous() for f in backbone(img)]
    t_bb = time.time() - t0
    feats64 = [f.contiguous() for f in copy.deepcopy(backbone).double()(img.double())]
    bb_noise = max(float((a.double() - b).abs().max() / b.abs().max()) for a, b in zip(feats, feats64))
    cfg = ref_shim.reference_head_cfg()
    cfg.pop("type", None)
    cfg["in_channels"] = list(chans)
    cfg["num_obj_query"] = num_obj_query
    head = ref_shim.build_reference_head(dict(cfg))
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in head.state_dict().items())
    sd0 = seeded.seeded_state_dict(shapes, WEIGHT_SEED)
    metas = [dict(img_shape=(H, W, 3), scale_factor=[sf] * 4)] * bs
    ops, rep = separate(OracleCrossHead2, dict(cfg), sd0, feats, metas, feats64=feats64)
    sd = seeded.apply_ops(dict(sd0), ops)
    head.load_state_dict(sd, strict=True)
    cls, masks, idx, dt = _run_e2e(head, feats, metas)
    o64 = OracleCrossHead2(**cfg).eval()
    o64.load_state_dict(sd)
    # the fp64 evaluation starts from the IMAGE (fp64 backbone features): the recorded list
    # must survive the backbone's own rounding as well
    gap, noise = _sep_check(o64.double(), head, feats64, metas, cls, idx, head.num_rel_query)
    print("%s: backbone %.1f s, fp32-vs-fp64 feature error %.2e; min gap %.3e, fp32-vs-fp64 "
          "importance error %.3e (margin %.0f)" % (name, t_bb, bb_noise, gap, noise, gap / noise))
    with torch.no_grad():
        res = head.get_bboxes(cls, masks, metas)
    m = masks["mask"]
    probe = torch.from_numpy(np.random.default_rng(img_seed + 1).integers(0, m.numel(), 4096))
    sub = torch.div(idx, head.num_obj_query, rounding_mode="trunc")
    out = dict(weight_seed=WEIGHT_SEED, weight_crc=seeded.checksum(sd), img_seed=img_seed,
               img_crc=seeded.checksum([img]), backbone_seed=bb_seed,
               backbone_crc=seeded.checksum([v for k, v in bsd.items()
                                             if v.dtype == torch.float32]),
               height=H, width=W, batch=bs, img_scale=sf, num_obj_query=num_obj_query,
               rel=_np(cls["rel"]), cls=_np(cls["cls"]), importance=_np(cls["importance"]),
               sub=_np(cls["sub"]), obj=_np(cls["obj"]),
               topk_idx=_np(idx), sub_pos=_np(sub), obj_pos=_np(idx - sub * head.num_obj_query),
               min_gap=gap, fp64_noise=noise, backbone_fp64_noise=bb_noise,
               mask_probe_idx=_np(probe), mask_probe=_np(m.flatten()[probe]),
               mask_neg_frac=float((m < 0).float().mean()), ref_seconds=dt, **ops)
    for l, f in enumerate(feats):
        pi = torch.from_numpy(np.random.default_rng(img_seed + 10 + l).integers(0, f.numel(), 8192))
        out["feat%d_shape" % l] = np.array(f.shape)
        out["feat%d_probe_idx" % l] = _np(pi)
        out["feat%d_probe" % l] = _np(f.flatten()[pi])
        out["feat%d_absmax" % l] = float(f.abs().max())
        out["feat%d_mean" % l] = float(f.double().mean())
    for i, r in enumerate(res):
        out["res%d_labels" % i] = _np(r[1])
        out["res%d_rel_pairs" % i] = _np(r[2])
        out["res%d_r_dists" % i] = _np(r[7])
        out["res%d_pan_img" % i] = _np(r[4]).astype(np.int32)
        rows = np.arange(0, r[3].shape[0], 1 if H * W < 100000 else 10)   # (fixture size)
        out["res%d_masks_rows" % i] = rows
        out["res%d_masks" % i] = np.packbits(_np(r[3])[rows])
        out["res%d_masks_shape" % i] = np.array((len(rows),) + tuple(r[3].shape[1:]))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def gen_ppn_sep():
    from .head import OracleCrossHead2
    cfg = ref_shim.reference_head_cfg()
    cfg.pop("type", None)
    head = ref_shim.build_reference_head()
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in head.state_dict().items())
    sd0 = seeded.seeded_state_dict(shapes, WEIGHT_SEED)
    q = seeded.uniform(np.random.default_rng(23), (100, 2, 256), -2.0, 2.0)
    ops, rep = separate(OracleCrossHead2, dict(cfg), sd0, None, None, q_override=q)
    sd = seeded.apply_ops(dict(sd0), ops)
    head.load_state_dict(sd, strict=True)
    with torch.no_grad():  # pairnet_head.py:322-340 with the reference's modules
        s = F.normalize(head.sub_query_update(q).transpose(0, 1), p=2, dim=-1, eps=1e-12)
        o = F.normalize(head.obj_query_update(q).transpose(0, 1), p=2, dim=-1, eps=1e-12)
        raw = torch.matmul(s, o.transpose(1, 2))
        imp = head.update_importance(raw)
        _, idx = torch.topk(imp.flatten(-2, -1), k=head.num_rel_query)
        sub = torch.div(idx, head.num_obj_query, rounding_mode="trunc")
        obj = torch.remainder(idx, head.num_obj_query)
        i64 = _ppn_scores(sd, q, torch.float64)
    gap, noise = float(topk_gaps(imp, 100).min()), _top_noise(imp, i64, 100)
    assert gap >= 1e-4 and gap >= MIN_MARGIN * noise, (gap, noise)
    np.savez_compressed(os.path.join(OUT, "ppn_sep.npz"), weight_seed=WEIGHT_SEED,
                        weight_crc=seeded.checksum(sd), query_feat=_np(q),
                        importance_raw=_np(raw), importance=_np(imp), topk_idx=_np(idx),
                        sub_pos=_np(sub), obj_pos=_np(obj), min_gap=gap, fp64_noise=noise, **ops)


RES_NAMES = ("bboxes", "labels", "rel_pairs", "masks", "pan_img", "r_scores", "r_labels",
             "r_dists")


def stable_feat_seed(head, sd, first, bs, H, W, metas, margin=1.2e-4, tries=600):
    """First feature seed >= `first` whose run is not chaotic.  With random weights a
    resized mask logit can sit within fp32 noise of zero; the attention masks threshold it
    (pairnet_head.py:244-256 / the sibling heads' forward_head), and at 96x128 (12 keys on
    the coarsest level) one flipped bit moves the outputs by 0.1.  Every bilinear resize of
    the run is watched: a seed is accepted when the smallest |resized mask logit| over all
    layers is >= `margin` (the typical minimum over the ~5e5 logits of a run is 2e-5; 1.2e-4 is
    an order of magnitude above the ~1e-5 by which an fp32
    re-association moves these logits), so that no mask bit hangs on rounding."""
    real = F.interpolate
    for seed in range(first, first + tries):
        head.load_state_dict(sd, strict=True)
        feats = seeded.seeded_feats(seed, bs, H, W)
        mf_bias = calibrate_mask_bias(head, feats)
        seen = []

        def watch(x, *a, **k):
            y = real(x, *a, **k)
            if k.get("mode", None) == "bilinear" and y.shape[-2:] != tuple(feats[0].shape[-2:]):
                seen.append(float(y.abs().min()))
            return y
        F.interpolate = watch
        try:
            with torch.no_grad():
                head.forward(feats, metas)
        finally:
            F.interpolate = real
        print("feat seed %d: smallest |resized mask logit| %.2e over %d resizes"
              % (seed, min(seen), len(seen)))
        if min(seen) >= margin:
            return seed, feats, mf_bias
    raise RuntimeError("no stable seed found")


def gen_baseline_small():
    head = ref_shim.build_reference_baseline_head()
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in head.state_dict().items())
    sd = seeded.seeded_state_dict(shapes, WEIGHT_SEED + 1)
    H, W, bs = 96, 128, 2
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0, 2.0, 2.0, 2.0])] * bs
    feat_seed, feats, mf_bias = stable_feat_seed(head, sd, 71, bs, H, W, metas)
    with torch.no_grad():
        cls, masks = head.forward(feats, metas)
        res = head.get_bboxes(cls, masks, metas)
    m = masks["mask"]                                   # (9, bs, Q, h, w)
    probe = torch.from_numpy(np.random.default_rng(72).integers(0, m[0].numel(), 4096))
    top2 = lambda x: (lambda v: (v[..., 0] - v[..., 1]).min())(x.topk(2, dim=-1)[0])
    fg = F.softmax(cls["rel"], -1)[..., 1:].flatten(1)
    out = dict(weight_seed=WEIGHT_SEED + 1, weight_crc=seeded.checksum(sd), feat_seed=feat_seed,
               feat_crc=seeded.checksum(feats), height=H, width=W, batch=bs,
               mask_last=_np(m[-1]), mask_probe_idx=_np(probe),
               mask_probe=_np(m.flatten(1)[:, probe]),
               sub_ids=_np(cls["subject_scores"].max(-1)[1]),
               obj_ids=_np(cls["object_scores"].max(-1)[1]),
               match_gap=float(min(top2(cls["subject_scores"]), top2(cls["object_scores"]))),
               rank_gap=_np(topk_gaps(fg.unsqueeze(1), 100)),
               mask_neg_frac=float((m < 0).float().mean()))
    out["override_" + MF_BIAS] = _np(mf_bias)
    for k, v in cls.items():
        out["cls_" + k] = _np(v)
    for i, r in enumerate(res):
        for name, v in zip(RES_NAMES, r):
            if name != "bboxes":                        # torch.rand dummies in the reference
                out["res%d_%s" % (i, name)] = np.packbits(_np(v)) if name == "masks" else _np(v)
        out["res%d_masks_shape" % i] = np.array(r[3].shape)
    np.savez_compressed(os.path.join(OUT, "baseline_small.npz"), **out)


def gen_psgtr2_small():
    head = ref_shim.build_reference_psgtr2_head()
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in head.state_dict().items())
    sd = seeded.seeded_state_dict(shapes, WEIGHT_SEED + 2)
    head.load_state_dict(sd, strict=True)
    H, W = 96, 128
    feats = seeded.seeded_feats(81, 1, H, W)
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0, 2.0, 2.0, 2.0])]
    mf_bias = calibrate_mask_bias(head, feats)
    with torch.no_grad():
        cls, masks = head.forward(feats, metas)
        res = head.get_bboxes(cls, masks, metas)
    out = dict(weight_seed=WEIGHT_SEED + 2, weight_crc=seeded.checksum(sd), feat_seed=81,
               feat_crc=seeded.checksum(feats), height=H, width=W, batch=1)
    out["override_" + MF_BIAS] = _np(mf_bias)
    for k, v in cls.items():
        out["cls_" + k] = _np(v)
    for k, v in masks.items():
        out["mask_" + k] = _np(v)
    for name, v in zip(RES_NAMES, res[0]):
        out["res0_" + name] = np.packbits(_np(v)) if name == "masks" else _np(v)
    out["res0_masks_shape"] = np.array(res[0][3].shape)
    np.savez_compressed(os.path.join(OUT, "psgtr2_small.npz"), **out)


# ---- CrossHeadBBox (pairnet_bbox_head.py) ------------------------------------------------
# The path makes three index selections.  Which 300 proposals are taken only permutes the
# decoder queries, but the ORDER of the 100 kept queries lays out the importance matrix the
# Matrix Learner convolves, and the pair top-k picks the relation decoder's inputs: parity
# beyond "close" needs all three to be decided by more than fp32 rounding.  The class logits
# of the last layer are scaled (x4, an op) to spread the query ranking, each image's feature
# seed is the first one whose kept-query ranking is separated by BBOX_MARGIN x the
# fp32-vs-fp64 difference of the same scores, and the PPN is separated as for CrossHead2
# (`separate(..., q_override=)`).
BBOX_MARGIN = 10.0
BBOX_CLS_GAIN = 4.0


def _bbox_models():
    from .bbox_head import OracleCrossHeadBBox
    from .deformable_detr import ChannelMapper
    neck_cfg, cfg = ref_shim.reference_bbox_cfg()
    head = ref_shim.build_reference_bbox_head(cfg)
    cfg = dict(cfg)
    cfg.pop("type")
    oracle = OracleCrossHeadBBox(**cfg).eval()
    ncfg = dict(neck_cfg)
    ncfg.pop("type")
    neck = ChannelMapper(**ncfg).eval()
    sd = seeded.seeded_state_dict(
        OrderedDict((k, tuple(v.shape)) for k, v in head.state_dict().items()), WEIGHT_SEED + 2)
    nsd = seeded.seeded_state_dict(
        OrderedDict((k, tuple(v.shape)) for k, v in neck.state_dict().items()), WEIGHT_SEED + 3)
    neck.load_state_dict(nsd, strict=True)
    return head, oracle, neck, cfg, sd, nsd


def bbox_feats(seeds, H, W, smooth=0):
    """One image per seed (images of a batch are independent): C3..C5 of a ResNet; `smooth`:
    seeded.smooth_feats' coarsening factor (0: per-pixel white noise)."""
    per = [(seeded.smooth_feats(int(s), 1, H, W, smooth) if smooth
            else seeded.seeded_feats(int(s), 1, H, W))[1:] for s in seeds]
    return [torch.cat([p[l] for p in per], 0) for l in range(3)]


def _by_token(trace, key, n_tokens):
    """A per-query quantity scattered to its proposal's token index (NaN elsewhere): decoder
    queries are identified by the token their proposal came from, whatever the order of the
    proposal list."""
    v = trace[key]
    out = torch.full((v.shape[0], n_tokens), float("nan"), dtype=torch.float64)
    return out.scatter(1, trace["topk_proposals"], v.double())


def _kept_tokens(trace):
    return torch.gather(trace["topk_proposals"], 1, trace["index"])


def _keep_margin(t32, t64):
    """(min gap between consecutive kept-query scores, fp32-vs-fp64 difference of the scores of
    the 200 best queries compared token by token, same kept list?)"""
    n = int(max(t32["topk_proposals"].max(), t64["topk_proposals"].max())) + 1
    a, b = _by_token(t32, "query_score", n), _by_token(t64, "query_score", n)
    s, o = t32["query_score"].sort(dim=-1, descending=True)
    gap = float((s[..., :100] - s[..., 1:101]).min())
    top_tok = torch.gather(t32["topk_proposals"], 1, o[..., :200])
    d = (a - b).abs().gather(1, top_tok)
    nz = float(d.nan_to_num(nan=float("inf")).max())     # a top query missing in fp64: inf
    return gap, nz, torch.equal(_kept_tokens(t32), _kept_tokens(t64))


def gen_bbox(name, H, W, bs, first_seed, smooth=0):
    import copy
    head, oracle, neck, cfg, sd0, nsd = _bbox_models()
    ops = {"scale_cls_branches.5.weight": np.array([BBOX_CLS_GAIN, 0, cfg["num_classes"]])}
    sd = seeded.apply_ops(dict(sd0), ops)
    oracle.load_state_dict(sd, strict=True)
    o64, n64 = copy.deepcopy(oracle).double(), copy.deepcopy(neck).double()
    meta1 = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=[1.0] * 4)]
    seeds, seed = [], first_seed
    with torch.no_grad():
        while len(seeds) < bs:
            f = bbox_feats([seed], H, W, smooth)
            t32 = {}
            oracle(neck(f), meta1, trace=t32)
            s, o = t32["query_score"].sort(dim=-1, descending=True)
            gap = float((s[..., :100] - s[..., 1:101]).min())
            msg = "feat seed %d: kept-query gap %.2e" % (seed, gap)
            if gap >= 4e-6:            # (typical fp32-vs-fp64 difference: 0.3-3e-6)
                t64 = {}
                o64(n64([x.double() for x in f]), meta1, trace=t64)
                gap, nz, same = _keep_margin(t32, t64)
                msg += ", fp32-vs-fp64 %.2e -> margin %.1f" % (nz, gap / nz)
                if gap >= BBOX_MARGIN * nz and same:
                    seeds.append(seed)
            print(msg, flush=True)
            seed += 1
        feats = bbox_feats(seeds, H, W, smooth)
        sf = [float(W) / round(W / 1.6), float(H) / round(H / 1.6)] * 2
        metas = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=sf)] * bs
        nf = neck(feats)
        t32, t64 = {}, {}
        oracle(nf, metas, trace=t32)
        o64(n64([x.double() for x in feats]), metas, trace=t64)
        pops, rep = separate(None, cfg, sd, feats, metas,
                             q_override=(t32["query_feat"], t64["query_feat"]))
        ops.update(pops)
        sd = seeded.apply_ops(dict(sd0), ops)
        head.load_state_dict(sd, strict=True)
        oracle.load_state_dict(sd, strict=True)
        o64 = copy.deepcopy(oracle).double()
        t0 = time.time()
        cls, box = head(nf, metas)                       # the REFERENCE class
        dt = time.time() - t0
        res = head.get_bboxes(cls, box, metas, rescale=True)
        t32, t64 = {}, {}
        c2, b2 = oracle(nf, metas, trace=t32)
        assert all(torch.equal(cls[k], c2[k]) for k in cls) and all(torch.equal(box[k], b2[k]) for k in box)
        c64, _ = o64(n64([x.double() for x in feats]), metas, trace=t64)
    # margins of the three selections, against the fp32-vs-fp64 difference
    e0, e064 = t32["enc_cls0"], t64["enc_cls0"]
    P = t32["topk_proposals"].shape[1]
    s0 = e0.sort(dim=-1, descending=True)[0]
    prop_gap = float((s0[:, P - 1] - s0[:, P]).min())
    prop_noise = float((e0 - e064).abs().max())
    keep_gap, keep_noise, same_kept = _keep_margin(t32, t64)
    k = cfg["num_rel_query"]
    pair_gap = float(topk_gaps(cls["importance"], k).min())
    pair_noise = _top_noise(cls["importance"], c64["importance"], k)
    print("%s: proposals gap %.2e / %.2e, kept queries %.2e / %.2e, pairs %.2e / %.2e (gap / "
          "fp32-vs-fp64); reference forward %.1f s"
          % (name, prop_gap, prop_noise, keep_gap, keep_noise, pair_gap, pair_noise, dt))
    assert set(map(tuple, t32["topk_proposals"].sort(-1)[0].tolist())) == \
        set(map(tuple, t64["topk_proposals"].sort(-1)[0].tolist()))
    assert same_kept and torch.equal(t32["topk_idx"], t64["topk_idx"])
    assert prop_gap >= BBOX_MARGIN * prop_noise and keep_gap >= BBOX_MARGIN * keep_noise
    # (the pair scores inherit the trunk's rounding through queries that are 96 % common
    # component with random weights: 5 x is what a 1000-seed search reaches; the GPU test
    # asserts its own measured error against the gap)
    assert pair_gap >= 1e-4 and pair_gap >= 5.0 * pair_noise
    probe = torch.from_numpy(np.random.default_rng(91).integers(0, cls["enc_cls_scores"].numel(), 8192))
    bprobe = torch.from_numpy(np.random.default_rng(92).integers(0, box["bbox"].numel() * 0 + cls["enc_bbox_preds"].numel(), 4096))
    out = dict(weight_seed=WEIGHT_SEED + 2, neck_seed=WEIGHT_SEED + 3, weight_crc=seeded.checksum(sd0),
               neck_crc=seeded.checksum(nsd), feat_seeds=np.array(seeds), feat_crc=seeded.checksum(feats),
               height=H, width=W, batch=bs, img_scale=np.array(sf), feat_smooth=smooth,
               prop_gap=prop_gap, prop_noise=prop_noise, keep_gap=keep_gap, keep_noise=keep_noise,
               pair_gap=pair_gap, pair_noise=pair_noise,
               proposals=_np(t32["topk_proposals"]), keep_index=_np(t32["index"]),
               topk_idx=_np(t32["topk_idx"]), sub_pos=_np(t32["sub_pos"]), obj_pos=_np(t32["obj_pos"]),
               enc_probe_idx=_np(probe), enc_cls_probe=_np(cls["enc_cls_scores"].flatten()[probe]),
               box_probe_idx=_np(bprobe), enc_box_probe=_np(cls["enc_bbox_preds"].flatten()[bprobe]),
               query_score=_np(t32["query_score"]))
    for kk in ("sub", "obj", "cls", "rel", "importance"):
        out["cls_" + kk] = _np(cls[kk])
    for kk in box:
        out["bbox_" + kk] = _np(box[kk])
    for i, r in enumerate(res):
        out["res%d_det" % i], out["res%d_labels" % i] = _np(r[0]), _np(r[1])
        out["res%d_pairs" % i], out["res%d_r_dists" % i] = _np(r[2]), _np(r[5])
    out.update(ops)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    want = lambda n: args.only in (None, n)
    if want("convtiny"):
        gen_convtiny()
    if want("msda"):
        gen_msda()
    if any(want(n) for n in ("ppn", "reldec", "fwdhead", "declayer", "e2e_small", "e2e_full")):
        head, sd = build_ref()
        if want("ppn"):
            gen_ppn(head, sd)
        if want("reldec"):
            gen_reldec(head, sd)
        if want("fwdhead"):
            gen_fwdhead(head, sd)
        if want("declayer"):
            gen_declayer(head, sd)
        if want("e2e_small"):
            gen_e2e_small(head, sd)
        if want("e2e_full"):
            gen_e2e_full(head, sd)
    if want("ppn_sep"):
        gen_ppn_sep()
    if want("e2e_small_sep"):
        gen_e2e_sep("e2e_small_sep", 96, 128, 2, 53, 2.0, screen=6e-5)
    if want("e2e_full_sep"):
        gen_e2e_sep("e2e_full_sep", 800, 1333, 1, 63, 2.083)
    if want("e2e_image_full"):      # BASELINE configs[1]: R50, 100 queries, 800x1333
        gen_e2e_image("e2e_image_full", "r50", 800, 1333, 2, 163, 31, 2.083)
    if want("e2e_image_swinl"):     # BASELINE configs[3]: Swin-L (true dims), 200 queries
        gen_e2e_image("e2e_image_swinl", "swinL", 256, 320, 2, 173, 41, 1.0, num_obj_query=200)
    if want("e2e_image_swinl_full"):   # configs[3] at the production size, one image
        # (SEP_XATTN=0.03 for this one: at 21 950 keys x 200 queries x 9 layers a few attention-
        # mask bits differ between the fp32 and the fp64 evaluation, and with the default 0.1
        # the margin of the recorded list is 3 instead of >= 10; image seed: SWINL_FULL_SEED)
        gen_e2e_image("e2e_image_swinl_full", "swinL", 800, 1333, 1,
                      int(os.environ.get("SWINL_FULL_SEED", 183)), 41, 2.083, num_obj_query=200)
    if want("baseline_small"):
        gen_baseline_small()
    if want("psgtr2_small"):
        gen_psgtr2_small()
    if want("bbox_small"):
        gen_bbox("bbox_small", 160, 192, 2, 300)
    if want("bbox_full"):
        gen_bbox("bbox_full", 800, 1333, 1, 500, smooth=16)   # (see seeded.smooth_feats)
    for f in sorted(os.listdir(OUT)):
        print("%-16s %8.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))


if __name__ == "__main__":
    main()
