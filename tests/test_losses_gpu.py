"""GPU: `CrossHead2.loss` (pair-net_amd/losses.py + csrc/loss.hip) on the HIP head's own
outputs against oracle/losses.py (pinned bit for bit to the reference's methods,
tests/test_losses.py) on the same tensors and the same sampled points.  Tolerance: 1e-4
relative on every loss value (fp32 sums in a different order); the two Hungarian assignments
must be identical."""
import numpy as np
import pytest
import torch

from helpers import head_cfg
from oracle import mmdet_train as T
from oracle.losses import OracleCrossHead2Loss

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _outputs(seed, H=96, W=128, bs=2):
    from pairnet_amd import CrossHead2
    head = CrossHead2(**head_cfg())
    head.init_weights(seed=3)
    head.to(DEV)
    g = torch.Generator().manual_seed(seed)
    feats = [torch.randn(bs, c, H // s, W // s, generator=g).to(DEV)
             for c, s in zip((256, 512, 1024, 2048), (4, 8, 16, 32))]
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0] * 4)] * bs
    cls, masks = head.forward(feats, metas)
    gt_labels = [torch.tensor([3, 17, 90, 120, 3]), torch.tensor([5, 60, 7])][:bs]
    gt_masks = [torch.rand(5, H, W, generator=g) > 0.6, torch.rand(3, H, W, generator=g) > 0.5][:bs]
    gt_rels = [torch.tensor([[0, 1, 5], [2, 3, 17], [1, 0, 56], [4, 2, 5], [0, 1, 9]]),
               torch.tensor([[0, 1, 2], [2, 1, 30]])][:bs]
    pts = [torch.rand(1, 12544, 2, generator=g) for _ in range(bs)]
    return head, cls, masks, metas, gt_rels, gt_labels, gt_masks, pts


@pytest.mark.parametrize("seed", [1, 2])
def test_loss_values_and_assignments_match_the_oracle(seed):
    head, cls, masks, metas, gt_rels, gt_labels, gt_masks, pts = _outputs(seed)
    oracle = OracleCrossHead2Loss()
    cpu = lambda d: {k: v.detach().cpu().clone() for k, v in d.items()}
    otrace = {}
    want = oracle.loss(cpu(cls), cpu(masks), gt_rels, gt_labels, gt_masks, point_coords=pts,
                       trace=otrace)
    trace = []
    got = head.loss(cls, masks, gt_rels, None, gt_labels, gt_masks, metas, point_coords=pts,
                    trace=trace)
    assert set(got) == set(want)
    for i, (tr, im) in enumerate(zip(trace, otrace["images"])):
        # sampled points of predictions and ground truths (bilinear, zero padding)
        assert float((tr["pred_pts"].cpu() - im["pred_pts"]).abs().max()) < 1e-5
        assert float((tr["gt_pts"].cpu() - im["gt_pts"]).abs().max()) < 1e-6
        # Hungarian assignments: queries <-> ground-truth objects, relation queries <-> triplets
        ref_gt = im["mask_gt_inds"].numpy()
        mine = np.zeros_like(ref_gt)
        mine[tr["mask_rows"]] = tr["mask_cols"] + 1
        assert np.array_equal(mine, ref_gt), i
        ref_tri = im["triplet_gt_inds"].numpy()
        mine = np.zeros_like(ref_tri)
        mine[tr["triplet_rows"]] = tr["triplet_cols"] + 1
        assert np.array_equal(mine, ref_tri), i
    errs = {k: abs(float(got[k]) - float(want[k])) / max(1.0, abs(float(want[k]))) for k in want}
    print("loss values", {k: float(v) for k, v in got.items()}, "relative errors", errs)
    assert all(np.isfinite(float(v)) and float(v) > 0 for v in got.values())
    assert all(e < 1e-4 for e in errs.values()), errs
    # SeesawLoss's persistent label counts advanced like the oracle's
    assert np.array_equal(head._loss.cum_samples, oracle.rel_loss.cum_samples.numpy())
    # a second batch weighs with the accumulated counts
    want2 = oracle.loss(cpu(cls), cpu(masks), gt_rels, gt_labels, gt_masks, point_coords=pts)
    got2 = head.loss(cls, masks, gt_rels, None, gt_labels, gt_masks, metas, point_coords=pts)
    for k in want2:
        assert abs(float(got2[k]) - float(want2[k])) < 1e-4 * max(1.0, abs(float(want2[k]))), k
    assert float(got2["loss_r_cls"]) != float(got["loss_r_cls"])


@pytest.mark.parametrize("seed", [1, 2])
def test_loss_gradients_match_autograd_through_the_oracle(seed):
    """SURVEY 8 f-4, first backward slice: d (loss_r_cls + loss_sub_cls + loss_obj_cls + loss_match)
    / d {rel, sub, obj, importance} from the analytic derivative kernels (csrc/loss.hip) against
    torch autograd through the reference-pinned loss oracle (the same assignments, the same
    SeesawLoss counts): 1e-4 relative to the largest gradient entry; masked rows are exactly 0."""
    head, cls, masks, metas, gt_rels, gt_labels, gt_masks, pts = _outputs(seed)
    oracle = OracleCrossHead2Loss()
    leaf = {k: v.detach().cpu().clone().requires_grad_(k in ("rel", "sub", "obj", "importance"))
            for k, v in cls.items()}
    want = oracle.loss(leaf, {k: v.detach().cpu().clone() for k, v in masks.items()}, gt_rels,
                       gt_labels, gt_masks, point_coords=pts)
    sum(want.values()).backward()
    grads = {}
    got = head.loss(cls, masks, gt_rels, None, gt_labels, gt_masks, metas, point_coords=pts,
                    grads=grads)
    assert set(grads) == {"rel", "sub", "obj", "importance"}
    for k in want:
        assert abs(float(got[k]) - float(want[k])) < 1e-4 * max(1.0, abs(float(want[k]))), k
    for k, g in grads.items():
        ref = leaf[k].grad
        assert tuple(g.shape) == tuple(ref.shape), k
        err = float((g.cpu() - ref).abs().max())
        scale = float(ref.abs().max())
        print(k, "max |grad|", scale, "max err", err)
        assert scale > 0 and err <= 1e-4 * scale, (k, err, scale)
        zero = ref.reshape(-1, ref.shape[-1]).abs().sum(-1) == 0
        if k != "importance":
            assert torch.all(g.cpu().reshape(-1, g.shape[-1])[zero] == 0), k


def test_match_cost_kernels_against_the_restated_costs():
    from pairnet_amd import hip
    g = torch.Generator().manual_seed(4)
    Q, G, Np = 100, 11, 12544
    cls, x = torch.randn(Q, 134, generator=g), torch.randn(Q, Np, generator=g) * 3
    t = torch.rand(G, Np, generator=g).round() * torch.rand(G, Np, generator=g)
    labels = torch.randint(0, 133, (G,), generator=g)
    want = T.ClassificationCost(2.0)(cls, labels) + T.CrossEntropyLossCost(5.0)(x, t) + \
        T.DiceCost(5.0, pred_act=True, eps=1.0)(x, t)
    cost = torch.empty(Q, G, device=DEV)
    hip.mask_match_cost(cls.to(DEV), labels.to(DEV), x.to(DEV), t.to(DEV), cost, 2.0, 5.0, 5.0, 1.0)
    assert float((cost.cpu() - want).abs().max()) < 2e-5 * float(want.abs().max())
    sub, obj, rel = (torch.randn(100, n, generator=g) for n in (134, 134, 56))
    gs, go, gr = (torch.randint(0, n, (G,), generator=g) for n in (133, 133, 56))
    want = T.ClassificationCost(1.0)(sub, gs) + T.ClassificationCost(1.0)(obj, go) + \
        T.ClassificationCost(0.5)(rel, gr)
    cost = torch.empty(100, G, device=DEV)
    hip.id_match_cost(sub.to(DEV), obj.to(DEV), rel.to(DEV), gs.to(DEV), go.to(DEV), gr.to(DEV),
                      cost, 1.0, 1.0, 0.5)
    assert float((cost.cpu() - want).abs().max()) < 1e-6


def test_point_sample_kernel_is_grid_sample():
    from pairnet_amd import hip
    g = torch.Generator().manual_seed(6)
    maps = torch.randn(7, 25, 42, generator=g)
    pts = torch.rand(500, 2, generator=g) * 1.2 - 0.1        # some points outside the map
    want = T.point_sample(maps.unsqueeze(1), pts.unsqueeze(0).repeat(7, 1, 1)).squeeze(1)
    out = torch.empty(7, 500, device=DEV)
    hip.point_sample(maps.to(DEV), pts.to(DEV), out)
    assert float((out.cpu() - want).abs().max()) < 1e-6
    masks = torch.rand(3, 30, 40, generator=g) > 0.5
    want = T.point_sample(masks.unsqueeze(1).float(), pts.unsqueeze(0).repeat(3, 1, 1)).squeeze(1)
    out = torch.empty(3, 500, device=DEV)
    hip.point_sample(masks.to(DEV), pts.to(DEV), out)
    assert float((out.cpu() - want).abs().max()) < 1e-6


def test_loss_refuses_an_image_without_relations():
    head, cls, masks, metas, gt_rels, gt_labels, gt_masks, pts = _outputs(3, bs=1)
    with pytest.raises(ValueError):
        head.loss(cls, masks, [torch.zeros(0, 3)], None, gt_labels, gt_masks, metas)
    with pytest.raises(NotImplementedError):
        head.forward_train()


@pytest.mark.parametrize("h,w,H,W", [(90, 120, 96, 128), (96, 128, 96, 128), (61, 45, 75, 51),
                                      (800, 1216, 800, 1344)])
def test_ground_truth_mask_preparation_kernel_is_the_oracle_bit_for_bit(h, w, H, W):
    from oracle.losses import prepare_gt_masks
    from pairnet_amd import hip
    g = torch.Generator().manual_seed(h + w)
    mask = torch.rand(3, h, w, generator=g) > 0.5
    out = torch.empty((3, H // 2, W // 2), dtype=torch.uint8, device=DEV)
    with torch.cuda.device(DEV):
        hip.gt_mask_prepare(mask.to(DEV), out, H, W)
    assert np.array_equal(out.cpu().numpy(), prepare_gt_masks(mask.numpy().astype(np.uint8), H, W))


def test_detector_val_losses_is_forward_train_without_the_backward():
    """PSGTr.val_losses (psgtr.py:113-146): backbone -> masks padded / resized -> head forward
    -> loss, against the oracle loss on the same head outputs and the oracle's prepared masks."""
    from oracle.losses import prepare_gt_masks
    from pairnet_amd import PSGTr
    from pairnet_amd.backbone import ResNet50Hip
    from pairnet_amd import CrossHead2
    H, W = 96, 128
    head = CrossHead2(**head_cfg())
    head.init_weights(seed=3)
    det = PSGTr.from_parts(ResNet50Hip(depth=50), head).to(DEV)
    g = torch.Generator().manual_seed(11)
    img = torch.randn(2, 3, H, W, generator=g).to(DEV)
    metas = [dict(img_shape=(90, 120, 3), scale_factor=[2.0] * 4, batch_input_shape=(H, W))] * 2
    gt_labels = [torch.tensor([3, 17, 90, 120, 3]), torch.tensor([5, 60, 7])]
    raw = [(torch.rand(5, 90, 120, generator=g) > 0.6).numpy().astype(np.uint8),
           (torch.rand(3, 90, 120, generator=g) > 0.5).numpy().astype(np.uint8)]

    class Bitmap:                       # mmdet BitmapMasks, as far as forward_train reads it
        def __init__(self, a):
            self.a = a

        def to_ndarray(self):
            return self.a
    gt_rels = [torch.tensor([[0, 1, 5], [2, 3, 17], [1, 0, 56], [4, 2, 5]]),
               torch.tensor([[0, 1, 2], [2, 1, 30]])]
    pts = [torch.rand(1, 12544, 2, generator=g) for _ in range(2)]
    got = det.val_losses(img, metas, gt_rels, None, gt_labels, [Bitmap(a) for a in raw],
                         point_coords=pts)
    cls, masks = head.forward(det.extract_feat(img), metas)
    cpu = lambda d: {k: v.detach().cpu().clone() for k, v in d.items()}
    prepared = [torch.from_numpy(prepare_gt_masks(a, H, W)) for a in raw]
    # (half the batch tensor; the mask logits are at a quarter: point sampling works in
    # normalised coordinates, the two grids never have to agree)
    assert tuple(prepared[0].shape[1:]) == (H // 2, W // 2)
    want = OracleCrossHead2Loss().loss(cpu(cls), cpu(masks), gt_rels, gt_labels, prepared,
                                       point_coords=pts)
    for k in want:
        assert abs(float(got[k]) - float(want[k])) < 1e-4 * max(1.0, abs(float(want[k]))), k
    with pytest.raises(NotImplementedError):
        det.forward(img, metas, return_loss=True)
    with pytest.raises(ValueError):
        det.val_losses(img, metas, gt_rels, None, gt_labels,
                       [np.zeros((5, H + 1, W), np.uint8), raw[1]])
