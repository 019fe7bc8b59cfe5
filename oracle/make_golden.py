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


RES_NAMES = ("bboxes", "labels", "rel_pairs", "masks", "pan_img", "r_scores", "r_labels",
             "r_dists")


def gen_baseline_small():
    head = ref_shim.build_reference_baseline_head()
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in head.state_dict().items())
    sd = seeded.seeded_state_dict(shapes, WEIGHT_SEED + 1)
    head.load_state_dict(sd, strict=True)
    H, W, bs = 96, 128, 2
    feats = seeded.seeded_feats(71, bs, H, W)
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0, 2.0, 2.0, 2.0])] * bs
    mf_bias = calibrate_mask_bias(head, feats)
    with torch.no_grad():
        cls, masks = head.forward(feats, metas)
        res = head.get_bboxes(cls, masks, metas)
    m = masks["mask"]                                   # (9, bs, Q, h, w)
    probe = torch.from_numpy(np.random.default_rng(72).integers(0, m[0].numel(), 4096))
    top2 = lambda x: (lambda v: (v[..., 0] - v[..., 1]).min())(x.topk(2, dim=-1)[0])
    fg = F.softmax(cls["rel"], -1)[..., 1:].flatten(1)
    out = dict(weight_seed=WEIGHT_SEED + 1, weight_crc=seeded.checksum(sd), feat_seed=71,
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
    if want("baseline_small"):
        gen_baseline_small()
    if want("psgtr2_small"):
        gen_psgtr2_small()
    for f in sorted(os.listdir(OUT)):
        print("%-16s %8.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))


if __name__ == "__main__":
    main()
