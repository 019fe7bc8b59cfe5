"""GPU: the sibling head `PSGTrHead2` (reference relation_heads/psgtr_head2.py) on the shared
trunk, against the golden vectors recorded from the reference class and the CPU oracle.
Tolerance 1e-3 on logits (fp32, north_star); labels exact wherever the reference's
softmax separates the two best classes."""
import numpy as np
import pytest
import torch

from helpers import golden, oracle_psgtr2_head, overrides_of, psgtr2_cfg
from oracle import seeded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = ("bboxes", "labels", "rel_pairs", "masks", "pan_img", "r_scores", "r_labels", "r_dists")


def _hip_head(sd):
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from pairnet_amd import PSGTrHead2
    head = PSGTrHead2(**psgtr2_cfg())
    head.load_state_dict(sd)
    return head.to(DEV)


def _err(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


def test_psgtr2_against_reference_golden():
    fx = golden("psgtr2_small")
    head_o, sd, crc = oracle_psgtr2_head(int(fx["weight_seed"]), overrides_of(fx))
    assert crc == int(fx["weight_crc"])
    H, W = int(fx["height"]), int(fx["width"])
    feats = seeded.seeded_feats(int(fx["feat_seed"]), 1, H, W)
    assert seeded.checksum(feats) == int(fx["feat_crc"])
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0, 2.0, 2.0, 2.0])]
    head = _hip_head(sd)
    cls, masks = head.forward([f.to(DEV) for f in feats], metas)
    torch.cuda.synchronize()
    assert set(cls) == {"sub", "obj", "rel"} and set(masks) == {"sub_seg", "obj_seg"}
    errs = {}
    for k in cls:
        assert tuple(cls[k].shape) == fx["cls_" + k].shape
        errs[k] = _err(cls[k], fx["cls_" + k])
    for k in masks:
        assert tuple(masks[k].shape) == fx["mask_" + k].shape
        errs[k] = _err(masks[k], fx["mask_" + k]) / max(1.0, float(np.abs(fx["mask_" + k]).max()))
    print("psgtr2_small errors:", errs)
    assert all(v < 1e-3 for v in errs.values()), errs
    res = head.get_bboxes(cls, masks, metas)
    torch.cuda.synchronize()
    r = res[0]
    # labels: exact where the reference's top-2 class probabilities are separated
    for half, key in ((slice(0, 100), "cls_sub"), (slice(100, 200), "cls_obj")):
        p = torch.softmax(torch.from_numpy(fx[key][0, 0]), -1)[:, :-1]
        top2 = p.topk(2, -1)[0]
        sure = ((top2[:, 0] - top2[:, 1]) > 1e-5).numpy()
        assert np.array_equal(r[1].cpu().numpy()[half][sure], fx["res0_labels"][half][sure])
    assert _err(r[7], fx["res0_r_dists"]) < 1e-5
    shape = tuple(fx["res0_masks_shape"])
    ref_masks = np.unpackbits(fx["res0_masks"])[:int(np.prod(shape))].reshape(shape).astype(bool)
    got = r[3].cpu().numpy()
    assert got.shape == shape and got.dtype == np.bool_
    assert (got != ref_masks).mean() < 1e-3
    assert np.array_equal(r[4].cpu().numpy(), fx["res0_pan_img"])
    assert np.array_equal(r[2].numpy(), fx["res0_rel_pairs"])
    for i in (0, 5, 6):
        assert tuple(r[i].shape) == fx["res0_" + NAMES[i]].shape and float(r[i].abs().sum()) == 0


def test_psgtr2_object_mask_is_the_initial_one():
    """The reference returns the object masks of the first forward_head call
    (psgtr_head2.py:404-411): obj_seg must not depend on the decoder layers' weights, and
    sub_mask_embed must not influence any output."""
    head_o, sd, _ = oracle_psgtr2_head(17)
    H, W = 64, 96
    feats = [f.to(DEV) for f in seeded.seeded_feats(18, 1, H, W)]
    metas = [dict(img_shape=(H, W, 3), scale_factor=[1.0] * 4)]
    head = _hip_head(sd)
    _, m1 = head.forward(feats, metas)
    obj1, sub1 = m1["obj_seg"].clone(), m1["sub_seg"].clone()
    sd2 = dict(sd)
    for k in sd:
        if k.startswith("transformer_decoder.layers."):
            sd2[k] = sd[k] * 1.5
    head2 = _hip_head(sd2)
    _, m2 = head2.forward(feats, metas)
    torch.cuda.synchronize()
    assert torch.equal(obj1, m2["obj_seg"])
    assert not torch.equal(sub1, m2["sub_seg"])
    sd3 = dict(sd)
    for k in sd:
        if k.startswith("sub_mask_embed."):
            sd3[k] = sd[k] * 0.0
    _, m3 = _hip_head(sd3).forward(feats, metas)
    torch.cuda.synchronize()
    assert torch.equal(sub1, m3["sub_seg"]) and torch.equal(obj1, m3["obj_seg"])


def test_psgtr2_batch_graphs_pipeline():
    """Batch 2 (beyond the reference's get_bboxes) == two single-image calls; hipGraph
    replay through the 3-deep pipeline is bitwise the eager result."""
    from pairnet_amd import PipelinedHead
    head_o, sd, _ = oracle_psgtr2_head(23)
    H, W = 64, 96
    f2 = [f.to(DEV) for f in seeded.seeded_feats(24, 2, H, W)]
    metas = [dict(img_shape=(H, W, 3), scale_factor=[1.0] * 4)] * 2
    head = _hip_head(sd)
    both = [[t.clone() if t.is_cuda else t for t in r] for r in head.simple_test_bboxes(f2, metas)]
    for i in range(2):
        one = head.simple_test_bboxes([f[i:i + 1].contiguous() for f in f2], metas[:1])[0]
        for name, x, y in zip(NAMES, both[i], one):
            if name == "r_dists":
                assert _err(x, y.cpu()) < 1e-5
            elif name == "masks":
                assert (x.cpu() != y.cpu()).float().mean() < 1e-3
            elif name == "labels":
                assert (x.cpu() != y.cpu()).float().mean() < 0.02
            else:
                assert torch.equal(x.cpu(), y.cpu())
    head.use_graphs = True
    pipe = PipelinedHead(head, depth=3)
    outs = []
    for _ in range(4):
        o = pipe.submit(f2, metas)
        if o is not None:
            outs.append([[t.clone() if t.is_cuda else t for t in r] for r in o])
    outs += pipe.flush()
    torch.cuda.synchronize()
    assert len(outs) == 4
    for o in outs:
        for ra, rb in zip(both, o):
            for x, y in zip(ra, rb):
                assert torch.equal(x.cpu(), y.cpu())


def test_psgtr2_at_800x1333_against_the_oracle():
    """The sibling at the production shape (its golden vectors are 96 x 128): every output
    of forward() against the CPU oracle (pinned to the reference class on the small fixture)."""
    head_o, sd, _ = oracle_psgtr2_head(29)
    H, W = 800, 1333
    feats = seeded.seeded_feats(30, 1, H, W)
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.083] * 4)]
    head = _hip_head(sd)
    with torch.no_grad():
        cls_o, masks_o = head_o.forward(feats, metas)
    cls, masks = head.forward([f.to(DEV) for f in feats], metas)
    torch.cuda.synchronize()
    errs = {k: _err(cls[k], cls_o[k]) for k in cls}
    for k in masks:
        errs[k] = _err(masks[k], masks_o[k]) / max(1.0, float(masks_o[k].abs().max()))
    print("psgtr2 800x1333 errors:", errs)
    assert all(v < 1e-3 for v in errs.values()), errs
