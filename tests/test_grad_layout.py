"""CPU: the flat gradient layouts of pair-net_amd/grad.py cover exactly the parameters the
reference's loss reaches -- decided by torch autograd through the reference-pinned oracle head, not
by a hand-written list.  (The gradients' VALUES are GPU tests: tests/test_grad_gpu.py.)"""
import torch

from helpers import oracle_head
from oracle import seeded
from oracle.head import OracleCrossHead2


def _reached_parameters():
    """Names of the oracle head's parameters with a non-zero gradient of <rel, G1> + <importance,
    G2> -- the two logits through which the reference's four loss terms reach the network
    (`loss_sub_cls` / `loss_obj_cls` read detached class logits, pairnet_head.py:380-390)."""
    head_o, _, _ = oracle_head(1234)
    for p in head_o.parameters():
        p.requires_grad_(True)
    H, W = 64, 96
    feats = seeded.seeded_feats(99, 1, H, W)
    metas = [dict(img_shape=(H, W, 3), scale_factor=[1.5] * 4)]
    trace = {}
    cls, _ = OracleCrossHead2.forward.__wrapped__(head_o, feats, metas, trace)
    # (the oracle evaluates the relation decoder under no_grad: re-evaluate it with the graph on)
    _, rel = OracleCrossHead2.relation_logits.__wrapped__(head_o, trace["query_feat"],
                                                          trace["sub_pos"], trace["obj_pos"])
    g = torch.Generator().manual_seed(3)
    loss = (rel * torch.randn(rel.shape, generator=g)).sum() + \
        (cls["importance"] * torch.randn(cls["importance"].shape, generator=g)).sum()
    loss.backward()
    return {n for n, p in head_o.named_parameters()
            if p.grad is not None and float(p.grad.abs().max()) > 0}, \
        {n for n, _ in head_o.named_parameters()}


def test_layouts_are_the_parameters_autograd_reaches():
    from pairnet_amd import CrossHead2, HeadGrad, PixelDecoderGrad, RelationTailGrad
    from helpers import head_cfg
    head = CrossHead2(**head_cfg())
    reached, every = _reached_parameters()
    names = lambda cls: [n for g, ns in cls.param_groups(head) if g != "cls" for n in ns]
    covered = set(names(HeadGrad)) | set(names(PixelDecoderGrad))
    assert covered == reached, (sorted(covered - reached)[:5], sorted(reached - covered)[:5])
    assert set(names(RelationTailGrad)) <= set(names(HeadGrad))
    # what the loss does NOT reach: the mask branch, the class head, the dead weights
    untouched = every - reached
    for n in ("cls_embed.weight", "mask_embed.0.weight", "pixel_decoder.mask_feature.weight",
              "pixel_decoder.lateral_convs.0.conv.weight", "pixel_decoder.output_convs.0.conv.weight",
              "transformer_decoder.post_norm.weight", "rel_query_embed3.weight"):
        assert n in untouched, n
    # no name twice, and every layout's size is the padded sum of its tensors
    for cls in (RelationTailGrad, HeadGrad, PixelDecoderGrad):
        all_names = [n for _, ns in cls.param_groups(head) for n in ns]
        assert len(all_names) == len(set(all_names))
        want = sum((head._params[n].numel() + 63) // 64 * 64 for n in all_names)
        assert cls.size_of(head) == want


def test_backbone_layout_is_stages_2_to_4_convolutions():
    from pairnet_amd import BackboneGrad, ResNet50Hip
    bb = ResNet50Hip()
    names = [n for _, ns in BackboneGrad.param_groups(bb) for n in ns]
    want = [k for k in bb._params if k.endswith(".weight") and k.startswith(("layer2", "layer3", "layer4"))
            and ("conv" in k or "downsample.0" in k)]
    assert sorted(names) == sorted(want) and len(names) == 42
    # completion order: the deepest block first
    assert names[0].startswith("layer4.2.") and names[-1].startswith("layer2.0.")


def test_step_lr_is_the_reference_schedule():
    from pairnet_amd.train import step_lr
    lrs = [step_lr(1e-4, e) for e in range(15)]          # max_epochs = 15, step = [5, 10], gamma 0.5
    assert lrs[:5] == [1e-4] * 5 and lrs[5:10] == [5e-5] * 5 and lrs[10:] == [2.5e-5] * 5
