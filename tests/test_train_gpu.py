"""GPU: the optimizer kernels (csrc/optim.hip) against torch.optim.AdamW + clip_grad_norm_ -- the
reference's optimizer and OptimizerHook (configs/mask2former/pairnet.py:353-368) -- and the
training step of Pair-Net's own parameters (pair-net_amd/train.py) end to end."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_adamw_and_clip_kernels_equal_torch():
    """Three steps over a flat buffer of three segments (different lr / decay multipliers, one
    a zero-decay "norm" segment) with global-norm clipping and a data-parallel pre-scale: every
    parameter, both moments, the norm and the clip coefficient against torch on the host."""
    from pairnet_amd import hip
    g = torch.Generator().manual_seed(5)
    sizes, pads = [1000, 37, 4096], [1024, 64, 4096]
    lr_mult, wd_mult = [1.0, 0.1, 1.0], [1.0, 1.0, 0.0]
    offs = np.concatenate([[0], np.cumsum(pads)]).astype(np.int64)
    n = int(offs[-1])
    ref_p = [torch.randn(s, generator=g).double().requires_grad_() for s in sizes]
    lr, wd, b1, b2, eps, max_norm, pre = 1e-3, 1e-2, 0.9, 0.999, 1e-8, 0.1, 0.5
    opt = torch.optim.AdamW([dict(params=[p], lr=lr * lm, weight_decay=wd * wm)
                             for p, lm, wm in zip(ref_p, lr_mult, wd_mult)], betas=(b1, b2), eps=eps)
    flat_p = torch.zeros(n)
    for p, o in zip(ref_p, offs):
        flat_p[o:o + p.numel()] = p.detach().float()
    flat_p = flat_p.to(DEV)
    flat_m, flat_v = torch.zeros_like(flat_p), torch.zeros_like(flat_p)
    seg_off = torch.from_numpy(offs).to(DEV)
    seg_lr = torch.tensor(lr_mult, device=DEV)
    seg_wd = torch.tensor(wd_mult, device=DEV)
    clip = torch.zeros(2, device=DEV)
    scratch = torch.zeros(256, device=DEV, dtype=torch.float64)
    for step in range(1, 4):
        grads = [torch.randn(s, generator=g) * (10.0 if step == 2 else 0.001) for s in sizes]
        flat_g = torch.zeros(n)
        for gr, o in zip(grads, offs):
            flat_g[o:o + gr.numel()] = gr
        flat_g = flat_g.to(DEV)
        hip.grad_norm_clip(flat_g, clip, scratch, pre=pre, max_norm=max_norm)
        hip.adamw(flat_p, flat_g, flat_m, flat_v, seg_off, seg_lr, seg_wd, lr, b1, b2, eps, wd, step,
                  clip=clip, pre=pre)
        for p, gr in zip(ref_p, grads):
            p.grad = (gr.float() * pre).double()
        norm = torch.nn.utils.clip_grad_norm_(ref_p, max_norm)
        opt.step()
        assert abs(float(clip[0]) - float(norm)) < 1e-5 * float(norm)
        assert abs(float(clip[1]) - min(1.0, max_norm / (float(norm) + 1e-6))) < 1e-6
        got = flat_p.cpu()
        for p, o in zip(ref_p, offs):
            err = float((got[o:o + p.numel()].double() - p.detach()).abs().max())
            assert err < 2e-6, (step, err)
    # the padding between segments never moves
    assert float(flat_p[1000:1024].abs().max()) == 0.0


@pytest.mark.parametrize("scope", ["tail", "head", "head+pixel_decoder", "all"])
def test_tail_training_step_updates_like_adamw_and_keeps_inference_consistent(scope):
    """`TailTrainer.step` on a fixed batch: (1) the first update of every trained tensor equals
    torch's AdamW + clip on the gradients the step produced; (2) the loss the step optimises goes
    down over a few steps; (3) after training, the INFERENCE kernels (packed [V|Q|K] / [V|K]
    projections, the Matrix Learner's packed weights, the repeated initial queries) and the taped
    forward give the same outputs -- the derived weight packs were refreshed; (4) `write_back()`
    puts the trained values into `state_dict()`; the frozen detector did not move; a FRESH head
    loaded from that state dict gives bit for bit the trained head's outputs (every derived pack
    -- [V|Q|K], [V|K], ConvTiny layouts, repeated initial queries and their mask embedding, the
    key position tables that carry `level_embed`, the encoder's [value | offsets | weights]
    projection, its bf16-plane weight splits and the query position tables that carry
    `level_encoding` -- was refreshed).  `scope`: the tail alone; + the nine masked decoder layers;
    + the pixel decoder's encoder path (both at lr_mult 0.1 as the reference's `transformer_decoder`
    / `pixel_decoder` groups), i.e. everything the loss reaches behind the backbone; "all": + the
    ResNet-50's stages 2-4 from the IMAGE (BatchNorm, stem and layer1 frozen as in the reference's
    config) -- the whole of the reference's trainable graph."""
    train_decoder = scope != "tail"
    from pairnet_amd import RelationTailGrad, TailTrainer
    from test_losses_gpu import _outputs
    head, cls, masks, metas, gt_rels, gt_labels, gt_masks, pts = _outputs(2, H=96, W=128, bs=2)
    g = torch.Generator().manual_seed(2)
    feats = [torch.randn(2, c, 96 // s, 128 // s, generator=g).to(DEV)
             for c, s in zip((256, 512, 1024, 2048), (4, 8, 16, 32))]
    before = {k: v.clone() for k, v in head.state_dict().items()}
    lr = 1e-3
    bb = None
    if scope == "all":
        from pairnet_amd import ResNet50Hip
        bb = ResNet50Hip().to(DEV)
        bb_before = {k: v.clone() for k, v in bb.state_dict().items()}
        feats = torch.randn(2, 3, 96, 128, generator=g).to(DEV)       # the IMAGE
    tr = TailTrainer(head, lr=lr, train_decoder=train_decoder,
                     train_pixel_decoder=scope == "head+pixel_decoder", backbone=bb)
    p0 = tr.flat_p.clone()
    out = tr.step(feats, metas, gt_rels, gt_labels, gt_masks, point_coords=pts)
    torch.cuda.synchronize()
    # (1) the update, recomputed with torch from the step's own gradients
    gflat = tr.flat_grad.cpu().double()
    norm = float(gflat.norm())
    assert abs(float(out["grad_norm"]) - norm) < 1e-4 * norm and norm > 0
    coef = min(1.0, tr.max_norm / (norm + 1e-6))
    worst = 0.0
    for n, v in tr.params.items():
        o, shape, k = tr.layout[n]
        p = p0[o:o + k].cpu().double().requires_grad_()
        is_norm = ".norms." in n or ".gn." in n
        lr_n = lr * (0.1 if ("transformer_decoder" in n or "pixel_decoder" in n
                             or n.startswith("backbone.")) else 1.0)
        opt = torch.optim.AdamW([p], lr=lr_n, weight_decay=0.0 if is_norm else tr.wd,
                                betas=tr.betas, eps=tr.eps)
        p.grad = gflat[o:o + k] * coef
        opt.step()
        worst = max(worst, float((v.reshape(-1).cpu().double() - p.detach()).abs().max()))
    print("first AdamW update: worst |difference| vs torch %.2e (lr %.0e)" % (worst, lr))
    assert worst < 2e-2 * lr
    # (2) a few more steps on the same batch
    hist = [float(out["loss_match"])]
    for _ in range(7):
        hist.append(float(tr.step(feats, metas, gt_rels, gt_labels, gt_masks,
                                  point_coords=pts)["loss_match"]))
    print("loss_match over 8 steps:", ["%.4f" % h for h in hist])
    assert all(np.isfinite(h) for h in hist)
    if not train_decoder:     # (lr 1e-3 on one batch: bouncy; with the decoder's 14 M weights each
        # moving 1e-4 per Adam step without warm-up the first steps need not go down at all)
        assert hist[1] < hist[0] and min(hist) < 0.95 * hist[0]
    # (3) inference kernels vs the taped forward on the trained weights
    image = feats
    if bb is not None:
        feats = [f.clone(memory_format=torch.preserve_format) for f in bb(image)]
    outs, _ = head.forward(feats, metas)
    pl = head._last_plan
    taped = RelationTailGrad(head).forward(pl.q.clone(), pl.sub_pos, pl.obj_pos)
    torch.cuda.synchronize()
    for k in ("rel", "importance"):
        assert float((taped[k] - outs[k]).abs().max()) < 1e-4, k
    # (4) state dict
    tr.write_back()
    sd = head.state_dict()
    moved = [k for k in sd if not torch.equal(sd[k].cpu(), before[k].cpu())]
    head_names = [n for n in tr.names if not n.startswith("backbone.")]
    assert set(moved) == set(head_names), (set(moved) ^ set(head_names))
    for n in head_names:
        assert torch.equal(sd[n].cpu(), tr.params[n].cpu()), n
    outs2, _ = head.forward(feats, metas)          # (write_back must not trigger a stale re-pack)
    assert torch.equal(outs2["rel"], outs["rel"])
    from pairnet_amd import CrossHead2
    from helpers import head_cfg
    fresh = CrossHead2(**head_cfg())
    fresh.load_state_dict(sd)
    fresh.to(DEV)
    if bb is not None:         # ... and a fresh backbone from ITS written-back state dict
        from pairnet_amd import ResNet50Hip
        bsd = bb.state_dict()
        bmoved = {k for k in bsd if not torch.equal(bsd[k].cpu(), bb_before[k].cpu())}
        assert bmoved == {n[len("backbone."):] for n in tr.names if n.startswith("backbone.")}
        assert len(bmoved) == 42 and not any(k.startswith(("conv1", "bn1", "layer1")) for k in bmoved)
        assert torch.equal(bb(image)[3], feats[3])             # (no stale re-pack after write_back)
        fresh_bb = ResNet50Hip()
        fresh_bb.load_state_dict(bsd)
        fresh_bb.to(DEV)
        f2 = fresh_bb(image)
        for a, b_ in zip(f2, feats):
            assert torch.equal(a, b_)
    outs3, _ = fresh.forward(feats, metas)
    for k in ("rel", "importance", "cls"):
        assert torch.equal(outs3[k], outs2[k]), k
    assert any("transformer_decoder.layers" in n for n in tr.names) == train_decoder
    assert any("pixel_decoder.encoder" in n for n in tr.names) == (scope in ("head+pixel_decoder", "all"))
    # the class path is in the layout but frozen (no gradient in the reference's graph)
    assert "cls_embed.weight" in tr.layout and "cls_embed.weight" not in tr.names


def _run_workers(tmp_path, backend, world, extra_env=None, noapply=False, tag=""):
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    root = os.path.dirname(os.path.abspath(__file__))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra_env or {})
    outs = [str(tmp_path / ("%s%s_%d.pt" % (backend, tag, r))) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "ddp_worker.py"), str(r), str(world),
                               port, backend, outs[r]] + (["noapply"] if noapply else []),
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    logs = [p.communicate(timeout=400) for p in procs]
    for p, (so, se) in zip(procs, logs):
        assert p.returncode == 0, se[-3000:]
    return [torch.load(o) for o in outs]


def test_data_parallel_training_two_ranks_average_their_gradients(tmp_path):
    """Two ranks (gloo, sharing the test GPU), each with its own batch, one `TailTrainer` each
    (tail + masked decoder, 8 MiB buckets: four collectives per step, overlapped with the backward
    pass): the reduced gradient buffer equals, bit for bit, the sum of the two single-process
    gradients; both ranks apply `scale` = 1/2 and end two steps with identical parameters."""
    import os
    ddp = _run_workers(tmp_path, "gloo", 2)
    assert torch.equal(ddp[0]["params"], ddp[1]["params"])
    assert torch.equal(ddp[0]["grad1"], ddp[1]["grad1"])
    assert ddp[0]["scale"] == 0.5 and ddp[0]["buckets"] >= 3
    assert ddp[0]["collectives"] == 2 * ddp[0]["buckets"]
    singles = [_run_workers(tmp_path, "none", 1, {"DDP_BATCH_SEED": str(20 + r)}, noapply=True,
                            tag="_b%d" % r)[0] for r in range(2)]
    assert singles[0]["collectives"] == 0 and singles[0]["scale"] == 1.0
    want = singles[0]["grad1"] + singles[1]["grad1"]
    assert float(want.abs().max()) > 0
    assert torch.equal(ddp[0]["grad1"], want)


def test_data_parallel_training_through_rccl_with_one_rank(tmp_path):
    """The same step with a live RCCL communicator (world size 1, `force_collective`): every
    bucket goes through `all_reduce` on the side stream and comes back unchanged."""
    try:
        got = _run_workers(tmp_path, "nccl", 1, {"DDP_BATCH_SEED": "20"}, noapply=True)[0]
    except AssertionError as e:
        if "NCCL" in str(e).upper():
            pytest.skip("RCCL could not initialise here: " + str(e)[-300:])
        raise
    ref = _run_workers(tmp_path, "none", 1, {"DDP_BATCH_SEED": "20"}, noapply=True, tag="_ref")[0]
    assert got["collectives"] == got["buckets"] >= 3
    assert torch.equal(got["grad1"], ref["grad1"])


def test_detector_train_step_from_the_image():
    """`PSGTr.train_step` in `forward_train`'s argument order: ground-truth masks at image size are
    prepared like the reference's (pad + nearest half-size), the detector's own trainer runs the
    iteration over backbone stages 2-4 + head; stem / layer1 / BatchNorm and the mask branch stay."""
    from pairnet_amd import build_detector, pairnet_r50
    det = build_detector(pairnet_r50())
    det.bbox_head.init_weights(seed=4)
    det.to(DEV)
    g = torch.Generator().manual_seed(9)
    H, W = 96, 128
    img = torch.randn(1, 3, H, W, generator=g).to(DEV)
    metas = [dict(img_shape=(H, W, 3), scale_factor=[1.0] * 4, batch_input_shape=(H, W))]
    gt_labels = [torch.tensor([3, 17, 90, 120])]
    gt_masks = [(torch.rand(4, H, W, generator=g) > 0.6).numpy()]
    gt_rels = [torch.tensor([[0, 1, 5], [2, 3, 17], [1, 0, 56]])]
    bb0 = {k: v.clone() for k, v in det.backbone.state_dict().items()}
    hd0 = {k: v.clone() for k, v in det.bbox_head.state_dict().items()}
    for _ in range(2):
        out = det.train_step(img, metas, gt_rels, None, gt_labels, gt_masks)
    assert set(out) == {"loss_r_cls", "loss_sub_cls", "loss_obj_cls", "loss_match", "grad_norm"}
    assert all(np.isfinite(float(v)) for v in out.values()) and float(out["grad_norm"]) > 0
    det._trainer.write_back()
    bb1, hd1 = det.backbone.state_dict(), det.bbox_head.state_dict()
    moved = {k for k in bb1 if not torch.equal(bb1[k].cpu(), bb0[k].cpu())}
    assert len(moved) == 42 and all(k.startswith(("layer2", "layer3", "layer4")) for k in moved)
    hmoved = {k for k in hd1 if not torch.equal(hd1[k].cpu(), hd0[k].cpu())}
    assert "relation_decoder.layers.0.ffns.0.layers.1.weight" in hmoved
    assert "pixel_decoder.encoder.layers.0.attentions.0.sampling_offsets.weight" in hmoved
    assert not any(k.startswith(("mask_embed", "cls_embed", "pixel_decoder.mask_feature",
                                 "pixel_decoder.lateral_convs", "pixel_decoder.output_convs"))
                   for k in hmoved)
    res = det.simple_test(img, metas)              # inference still runs on the trained weights
    assert len(res) == 1
    # a state dict loaded behind the trainer's back re-packs the weights: the trainer must refuse
    stale = det._trainer
    det.bbox_head.load_state_dict(det.bbox_head.state_dict())
    det.simple_test(img, metas)
    with pytest.raises(RuntimeError, match="re-packed"):
        stale.step(img, metas, gt_rels, gt_labels, det._prepare_gt_masks(img, gt_masks))
    det._trainer = None                            # (train_step builds a fresh one)
    # mmdet's own form: what EpochBasedRunner.train calls per iteration
    rec = det.train_step(dict(img=img, img_metas=metas, gt_rels=gt_rels, gt_bboxes=None,
                              gt_labels=gt_labels, gt_masks=gt_masks), None)
    assert set(rec) == {"loss", "log_vars", "num_samples"} and rec["num_samples"] == 1
    terms = [v for k, v in rec["log_vars"].items() if "loss_" in k]
    assert len(terms) == 4 and abs(float(rec["loss"]) - sum(terms)) < 1e-3 * sum(terms)


def test_the_tail_overfits_a_fixed_batch_at_the_reference_learning_rate():
    """150 iterations of `TailTrainer.step` (tail scope, AdamW lr 1e-4 / weight decay 1e-4 / clip
    0.1: the reference's optimizer settings) on ONE batch of two images: the two loss terms that
    train -- `loss_match` (BCE on the importance matrix) and `loss_r_cls` (Seesaw on the relation
    logits) -- fall to a fraction of their initial values (measured: 6.94 -> 1.45 and 10.3 -> 0.41;
    `tools/dbg/overfit_probe.py`).  Gradients, optimizer and weight refresh work TOGETHER."""
    from pairnet_amd import TailTrainer
    from test_losses_gpu import _outputs
    head, cls, masks, metas, gt_rels, gt_labels, gt_masks, pts = _outputs(2, H=96, W=128, bs=2)
    g = torch.Generator().manual_seed(2)
    feats = [torch.randn(2, c, 96 // s, 128 // s, generator=g).to(DEV)
             for c, s in zip((256, 512, 1024, 2048), (4, 8, 16, 32))]
    tr = TailTrainer(head)
    first = last = None
    for i in range(150):
        out = tr.step(feats, metas, gt_rels, gt_labels, gt_masks, point_coords=pts)
        if i == 0:
            first = {k: float(v) for k, v in out.items()}
    last = {k: float(v) for k, v in out.items()}
    print("first", first, "last", last)
    assert last["loss_match"] < 0.5 * first["loss_match"]
    assert last["loss_r_cls"] < 0.3 * first["loss_r_cls"]
