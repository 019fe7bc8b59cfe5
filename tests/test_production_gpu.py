"""GPU: parity at the PRODUCTION shapes BASELINE.json names, from the image tensor.

  configs[1]  Pair-Net R50 + Mask2Former, 100 / 100 queries, bs = 1, 800x1333
  configs[2]  the same with 2 images per GPU
  configs[3]  Pair-Net Swin-L (embed 192, depths 2-2-18-2, heads 6-12-24-48, window 12)
              + Mask2Former, 200 object queries

Fixtures `e2e_image_full` / `e2e_image_swinl` (oracle/make_golden.py `gen_e2e_image`): a
seeded image -> the oracle backbone (pinned to HuggingFace transformers) -> the REFERENCE's
own CrossHead2 class (run from /root/reference under the shims), separated like the `*_sep`
fixtures so that strict top-k equality is a meaningful assertion; the generator also runs
the whole chain in fp64 FROM THE IMAGE and insists on the same pair list and a score margin
>= 10 x the fp32-vs-fp64 difference.  Here the image goes through `PSGTr.simple_test`
(psgtr.py:148-156: native backbone -- at 800x1333 with its split-K stage-3/4 layers and the
odd-sided 100x167 Winograd maps -- then head, get_bboxes, triplet2Result), eagerly and
through the bench's hipGraph + 4-stream schedule.

Tolerances: backbone features 1e-4 of their scale; relation / class logits 1e-3 (fp32,
BASELINE north_star); pair indices, labels, rel_pairs exact.
"""
import numpy as np
import pytest
import torch

from helpers import golden, overrides_of
from oracle import seeded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _err(a, b):
    return float((torch.as_tensor(a).detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


def _detector(fx, kind):
    """The detector the fixture was recorded for, with the fixture's weights."""
    from collections import OrderedDict
    from pairnet_amd import build_detector, pairnet_r50, pairnet_swin
    Q = int(fx["num_obj_query"])
    if kind == "r50":
        from oracle.backbone import seeded_backbone_state
        cfg = pairnet_r50()
        bsd = seeded_backbone_state(int(fx["backbone_seed"]))
    else:
        from oracle.swin import OracleSwin, seeded_swin_state
        cfg = pairnet_swin("L", num_obj_query=Q)
        b = cfg["backbone"]
        assert (b["embed_dims"], tuple(b["depths"]), tuple(b["num_heads"]), b["window_size"]) == \
            (192, (2, 2, 18, 2), (6, 12, 24, 48), 12)
        bsd = seeded_swin_state(OracleSwin(**{k: b[k] for k in ("embed_dims", "depths",
                                                                "num_heads", "window_size")}),
                                int(fx["backbone_seed"]))
    assert seeded.checksum([v for v in bsd.values() if v.dtype == torch.float32]) == \
        int(fx["backbone_crc"])
    det = build_detector(cfg)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in det.bbox_head.state_dict().items())
    sd = seeded.seeded_state_dict(shapes, int(fx["weight_seed"]))
    seeded.apply_ops(sd, overrides_of(fx))
    assert seeded.checksum(sd) == int(fx["weight_crc"])
    det.backbone.load_state_dict(bsd)
    det.bbox_head.load_state_dict(sd)
    return det.to(DEV)


def _image(fx):
    bs, H, W = int(fx["batch"]), int(fx["height"]), int(fx["width"])
    img = seeded.uniform(np.random.default_rng(int(fx["img_seed"])), (bs, 3, H, W), -2.0, 2.0)
    assert seeded.checksum([img]) == int(fx["img_crc"])
    sf = float(fx["img_scale"])
    metas = [dict(img_shape=(H, W, 3), scale_factor=[sf] * 4)] * bs
    return img, metas


def _check_features(fx, feats, images):
    for l, f in enumerate(feats):
        shape = tuple(fx["feat%d_shape" % l])
        assert tuple(f.shape[1:]) == shape[1:]
        full = torch.zeros(shape)          # probes index the recorded (batch, C, h, w) tensor
        full[list(images)] = f.detach().cpu().float()
        idx = torch.from_numpy(fx["feat%d_probe_idx" % l])
        per = int(np.prod(shape[1:]))
        keep = torch.isin(idx // per, torch.tensor(list(images)))
        e = _err(full.flatten()[idx[keep]], fx["feat%d_probe" % l][keep.numpy()])
        scale = float(fx["feat%d_absmax" % l])
        print("C%d %s: max err %.2e of scale %.1f (%d probes)" % (l + 2, shape, e, scale,
                                                                  int(keep.sum())))
        assert int(keep.sum()) > 1000 and e < 1e-4 * scale


def _check_head(fx, det, images, n_img):
    """Strict indices + logits of the head's last plan against the fixture rows `images`."""
    pl = det.bbox_head._last_plan
    rows = list(images)
    assert float(fx["min_gap"]) >= 1e-4 and float(fx["min_gap"]) >= 10 * float(fx["fp64_noise"])
    assert np.array_equal(pl.topk_idx.cpu().numpy(), fx["topk_idx"][rows])
    assert np.array_equal(pl.sub_pos.cpu().numpy(), fx["sub_pos"][rows])
    assert np.array_equal(pl.obj_pos.cpu().numpy(), fx["obj_pos"][rows])
    imp = pl.imp.cpu().numpy().reshape(n_img, -1)
    ref = fx["importance"][rows].reshape(n_img, -1)
    top = np.argsort(-ref, axis=1)[:, :200]
    e_top = float(np.abs(np.take_along_axis(imp - ref, top, 1)).max())
    errs = dict(rel=_err(pl.rel, fx["rel"][rows]), cls=_err(pl.cls, fx["cls"][rows]),
                sub=_err(pl.sub_cls, fx["sub"][rows]), obj=_err(pl.obj_cls, fx["obj"][rows]),
                importance=float(np.abs(imp - ref).max()) / max(1.0, float(np.abs(ref).max())))
    print("min gap %.3e, GPU score error on the top pairs %.3e (the reference's fp32-vs-fp64 "
          "error from the image: %.3e); errors %s" % (float(fx["min_gap"]), e_top,
                                                      float(fx["fp64_noise"]), errs))
    # two scores each off by at most e keep their order when the gap between them exceeds 2 e:
    # the sufficient condition for the strict equalities above (rounds 1-5 asked for gap / 4 with
    # nothing behind the 4; measured in round 6, bf16x3 encoder and no packed fp32: 0.26 of the gap
    # on the two-image R50 fixture)
    assert 2 * e_top < float(fx["min_gap"])
    assert all(e < 1e-3 for e in errs.values()), errs


# Mask / panoptic tolerances: 3-4 x what was MEASURED on MI355X in round 4 (gpurun_out/c2_prod.log
# and the round's final test run; printed by every run).  A mask bit differs where the upsampled
# logit lies within fp32 rounding of 0: measured 0.6e-6 ... 1.3e-6 of the mask bits
# at 800 x 1333 (R50: both images, one- and two-image launches; Swin-L / 200 queries) and
# 1.8e-6 / 2.1e-6 on the 256 x 320 Swin-L fixture (fewer bits per mask); the panoptic maps were
# identical everywhere (0 pixels) -- one flipped threshold can move one pixel, so a handful are
# allowed.  (Rounds 1-3 accepted 1e-3
# and 5e-3, derived from nothing.)
MASK_BIT_TOL = 7e-6
PAN_PIXEL_TOL = 2e-5


def _check_results(fx, results, images, Q):
    """`triplet2Result` fields (psgtr.py:15-51) against the reference's get_bboxes tuple."""
    H0 = round(int(fx["height"]) / float(fx["img_scale"]))
    W0 = round(int(fx["width"]) / float(fx["img_scale"]))
    for r, i in zip(results, images):
        assert isinstance(r.labels, np.ndarray) and isinstance(r.masks, np.ndarray)
        assert np.array_equal(r.labels, fx["res%d_labels" % i])
        assert np.array_equal(r.rel_pair_idxes, fx["res%d_rel_pairs" % i])
        assert r.rel_dists.shape == (100, 57) and _err(r.rel_dists, fx["res%d_r_dists" % i]) < 1e-3
        assert r.refine_bboxes.shape == (200, 5) and float(np.abs(r.refine_bboxes).sum()) == 0.0
        assert r.rel_labels.shape == (100,)
        assert r.masks.shape == (200, H0, W0) and r.masks.dtype == np.bool_
        shape = tuple(fx["res%d_masks_shape" % i])
        ref_masks = np.unpackbits(fx["res%d_masks" % i])[:int(np.prod(shape))].reshape(shape)
        got = r.masks[fx["res%d_masks_rows" % i]]
        mism = float((got != ref_masks.astype(bool)).mean())
        pan = float((r.pan_results != fx["res%d_pan_img" % i]).mean())
        print("image %d: mask bit mismatch %.2e, panoptic map mismatch %.2e" % (i, mism, pan))
        assert mism <= MASK_BIT_TOL and pan <= PAN_PIXEL_TOL, (mism, pan)
        assert r.formatted_masks["pan_results"] is r.pan_results
        assert r.pan_results.shape == (H0, W0)


@pytest.mark.parametrize("images", [(0,), (1,), (0, 1)])
def test_r50_800x1333_image_to_triplets_matches_the_reference(images):
    """configs[1] (one image) and configs[2] (two images per GPU) at 800x1333 through
    `PSGTr.simple_test`."""
    fx = golden("e2e_image_full")
    det = _detector(fx, "r50")
    img, metas = _image(fx)
    img = img[list(images)].contiguous().to(DEV)
    metas = metas[:len(images)]
    _check_features(fx, det.extract_feat(img), images)
    results = det.simple_test(img, metas)
    _check_head(fx, det, images, len(images))
    _check_results(fx, results, images, 100)


def test_r50_800x1333_graph_replay_and_pipeline_give_the_eager_result():
    """The bench's schedule (backbone + stages as hipGraph replays on the 4-stream pipeline,
    two stage-A streams) on the fixture's two images alternately: strict indices against the
    reference for every submission, results bitwise the eager ones."""
    from pairnet_amd import PipelinedHead
    fx = golden("e2e_image_full")
    det = _detector(fx, "r50")
    img, metas = _image(fx)
    imgs = [img[i:i + 1].contiguous().to(DEV) for i in range(2)]
    head, net = det.bbox_head, det.backbone
    eager = []
    for im in imgs:
        r = head.simple_test_bboxes(net(im), metas[:1])[0]
        eager.append([t.clone() for t in (r[1], r[7], r[4], head._last_plan.topk_idx)])
    head.use_graphs = net.use_graphs = True
    pipe = PipelinedHead(head, depth=4, a_streams=2)
    order = [0, 1, 1, 0, 0, 1, 0, 1, 1, 0, 1, 0]
    got = []
    # graphs are captured at quiet points only (plans.quiet, LABNOTES R5.9): one batch at a time
    # through every slot, twice (eager, then capturing), as PSGTr.warm_graphs does
    for _ in range(2 * pipe.depth):
        sl = pipe.count % len(pipe.streams_a)
        with torch.cuda.stream(pipe.streams_a[sl]):
            pipe.submit(net(imgs[0], slot=sl), metas[:1])
        torch.cuda.synchronize()
        pipe.flush()
        torch.cuda.synchronize()
    assert len(head._plans) == 4 and all(
        pl.graph_b is not None and any(e["graph"] is not None for e in pl.graphs_a.values())
        for pl in head._plans.values())
    assert all(pl.graph is not None for pl in net._plans.values())

    def take(res):
        # read on the chain stream that produced the results (ordered behind get_bboxes and in
        # front of the slot's next query chain, which runs on the same stream) and say so:
        # plan internals like topk_idx are slot buffers, not part of the returned tuple
        pl = head._last_plan
        with torch.cuda.stream(res.pipeline_stream):
            got.append([t.clone() for t in (res[0][1], res[0][7], res[0][4], pl.topk_idx)])
            pipe.consumed(res, res.pipeline_stream)
    for i in order:
        sl = pipe.count % len(pipe.streams_a)
        pipe.streams_a[sl].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(pipe.streams_a[sl]):
            res = pipe.submit(net(imgs[i], slot=sl), metas[:1])
            if res is not None:
                take(res)
    while pipe.queue:
        take(pipe._finish(pipe.queue.pop(0)))
    torch.cuda.synchronize()
    assert len(got) == len(order)
    for i, g in zip(order, got):
        assert np.array_equal(g[3].cpu().numpy(), fx["topk_idx"][i:i + 1])
        for a, b in zip(eager[i], g):
            assert torch.equal(a, b)


@pytest.mark.parametrize("name", ["e2e_image_swinl", "e2e_image_swinl_full"])
def test_swin_l_200_queries_image_to_triplets_matches_the_reference(name):
    """configs[3]: the TRUE Swin-L backbone (197 M parameters, 18-block third stage) under
    the 200-query head through `PSGTr.simple_test`, strict pair indices: two images at
    256 x 320 (`e2e_image_swinl`) and one image at the PRODUCTION size 800 x 1333
    (`e2e_image_swinl_full`, round 4: Swin-L's own widths -- K = 192 / 384 / 768 / 1536 split-K
    choices, 24 / 48-head windows -- at the size the 57.9 images/s figure is quoted on)."""
    fx = golden(name)
    det = _detector(fx, "swinL")
    img, metas = _image(fx)
    img = img.to(DEV)
    n = int(fx["batch"])
    images = tuple(range(n))
    assert (int(fx["height"]), int(fx["width"])) == ((800, 1333) if name.endswith("full") else (256, 320))
    _check_features(fx, det.extract_feat(img), images)
    results = det.simple_test(img, metas)
    assert det.bbox_head._last_plan.imp.shape == (n, 200, 200)
    _check_head(fx, det, images, n)
    _check_results(fx, results, images, 200)


def test_result_streamer_equals_simple_test():
    """`ResultStreamer` (pinned ring, copy stream, `PipelinedHead.consumed`) hands back the
    same `Result` fields as `PSGTr.simple_test` -> `triplet2Result`, for a different image in
    every step of the 4-stream pipeline."""
    from oracle.backbone import seeded_backbone_state
    from pairnet_amd import PipelinedHead, ResultStreamer, build_detector, pairnet_r50
    det = build_detector(pairnet_r50())
    det.backbone.load_state_dict(seeded_backbone_state(41))
    det.bbox_head.init_weights(seed=3)     # (the default init gives constant masks)
    det.to(DEV)
    H, W = 160, 224
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0] * 4)]
    g = torch.Generator().manual_seed(3)
    imgs = [torch.randn(1, 3, H, W, generator=g).to(DEV) for _ in range(7)]
    fields = ("refine_bboxes", "labels", "rel_pair_idxes", "rel_dists", "rel_labels",
              "pan_results", "masks")
    want = [det.simple_test(im, metas)[0] for im in imgs]
    want = [{k: np.array(getattr(r, k)) for k in fields} for r in want]
    assert 0.02 < want[0]["masks"].mean() < 0.98 and not np.array_equal(want[0]["masks"],
                                                                        want[1]["masks"])
    head, net = det.bbox_head, det.backbone
    # (the 200 x 80 x 112 bool masks travel as bits unless pack_masks=False)
    for graphs, pack in ((False, True), (True, True), (True, False)):
        head.use_graphs = net.use_graphs = graphs
        pipe = PipelinedHead(head, depth=4, a_streams=2)
        streamer = ResultStreamer(head, ring=3, pack_masks=pack)
        for rep in range(2):
            got = []

            def keep(results):
                # (the arrays are views of the ring entry: copy what outlives `ring` pushes)
                for r in results:
                    assert r.formatted_masks["pan_results"] is r.pan_results
                    got.append({k: np.array(getattr(r, k)) for k in fields})

            def take(res):
                if len(streamer) == streamer.ring:
                    keep(streamer.pop())
                streamer.push(res, pipe)
            for im in imgs:
                sl = pipe.count % len(pipe.streams_a)
                pipe.streams_a[sl].wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(pipe.streams_a[sl]):
                    res = pipe.submit(net(im, slot=sl), metas)
                    if res is not None:
                        take(res)
            with torch.cuda.stream(pipe.streams_a[0]):
                for res in pipe.flush():
                    take(res)
            while len(streamer):
                keep(streamer.pop())
            assert len(got) == len(want)
            for r, w in zip(got, want):
                for k in fields:
                    assert r[k].dtype == w[k].dtype and np.array_equal(r[k], w[k]), (graphs, pack, rep, k)
    with pytest.raises(RuntimeError):
        ResultStreamer(head, ring=1).pop()
    # without a pipeline: results of a plain simple_test_bboxes on the current stream, copies
    # on the streamer's own stream
    head.use_graphs = net.use_graphs = False
    streamer = ResultStreamer(head, ring=2)
    for im, w in zip(imgs[:3], want[:3]):
        streamer.push(head.simple_test_bboxes(net(im), metas))
        (r,) = streamer.pop()
        for k in fields:
            assert np.array_equal(np.array(getattr(r, k)), w[k]), k
    # two images per step: one ring entry carries both tuples
    two = torch.cat(imgs[:2], 0)
    streamer.push(head.simple_test_bboxes(net(two), metas * 2))
    got2 = streamer.pop()
    assert len(got2) == 2
    for r, w in zip(got2, want[:2]):
        for k in ("labels", "rel_pair_idxes", "rel_labels"):
            assert np.array_equal(np.array(getattr(r, k)), w[k]), k
        # (a two-image launch takes other split-K factors than two one-image launches: the
        # same logits to ~1e-6, so a mask / panoptic pixel on the threshold may differ)
        assert (np.array(r.masks) != w["masks"]).mean() < 1e-4
        assert (np.array(r.pan_results) != w["pan_results"]).mean() < 1e-3
        assert np.abs(np.array(r.rel_dists) - w["rel_dists"]).max() < 1e-5
    streamer.close()
    # images of different ORIGINAL sizes (keep-ratio evaluation sets): a ring entry's buffers
    # are sized by the largest shape seen and handed out as views -- after the largest image no
    # entry allocates again, and every image's fields still equal simple_test's
    streamer = ResultStreamer(head, ring=2)
    sfs = (1.0, 2.0, 1.6, 1.0, 2.0, 1.25)
    ptrs = []
    for j, sf in enumerate(sfs):
        m = [dict(img_shape=(H, W, 3), scale_factor=[sf] * 4)]
        w_ = det.simple_test(imgs[j % 3], m)[0]
        w_ = {k: np.array(getattr(w_, k)) for k in fields}
        streamer.push(head.simple_test_bboxes(net(imgs[j % 3]), m))
        (r,) = streamer.pop()
        assert r.masks.shape == (200, round(H / sf), round(W / sf))
        for k in fields:
            assert np.array_equal(np.array(getattr(r, k)), w_[k]), (sf, k)
        e = streamer.entries[j % 2]
        ptrs.append({n: b.data_ptr() for n, b in e["flat"].items()})
    # entry 0 saw its largest image first; entry 1 grew once (scale 2.0 -> 1.0), then stayed
    assert ptrs[2] == ptrs[0] and ptrs[4] == ptrs[0] and ptrs[3] != ptrs[1] and ptrs[5] == ptrs[3]
    streamer.close()


def test_two_heads_in_one_process_keep_their_own_grids():
    """a pipeline on one head (64 / 128 reserved workgroup slots) does not
    change another head's launches: its results stay bitwise what they were alone, and its
    `grid_reserve` stays 0."""
    from helpers import head_cfg, oracle_head
    from pairnet_amd import CrossHead2, PipelinedHead
    _, sd, _ = oracle_head(7)
    heads = []
    for _ in range(2):
        h = CrossHead2(**head_cfg())
        h.load_state_dict(sd)
        heads.append(h.to(DEV))
    a, b = heads
    H, W = 96, 128
    feats = [f.to(DEV) for f in seeded.seeded_feats(5, 1, H, W)]
    metas = [dict(img_shape=(H, W, 3), scale_factor=[1.0] * 4)]
    alone = [t.clone() for t in a.simple_test_bboxes(feats, metas)[0] if t.is_cuda]
    pipe = PipelinedHead(b, depth=3, a_streams=1, grid_trim=128)
    assert b.grid_reserve == 128 and a.grid_reserve == 0
    for _ in range(4):
        pipe.submit(feats, metas)
    with_b = pipe.flush()[-1][0]
    again = [t for t in a.simple_test_bboxes(feats, metas)[0] if t.is_cuda]
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(alone, again))
    # (the reserve changes which workgroup computes which tile, never a result)
    assert all(torch.equal(x, y) for x, y in zip(alone, [t for t in with_b if t.is_cuda]))


@pytest.mark.parametrize("H,W", [(160, 112), (101, 149), (224, 136), (1333, 800), (800, 1067)])
def test_keep_ratio_shapes_portrait_and_odd_sides(H, W):
    """The test pipeline's keep-ratio resize (configs/mask2former/pairnet.py:310-331) hands
    over portrait images and odd sides (mask feature 40x28 / 26x38 / 56x34), and at full size a
    portrait 1333 x 800 and a 4:3 800 x 1067 image (other tile counts for the stem, the
    Winograd transforms and every split-K decision than the 800 x 1333 fixtures): image ->
    native backbone -> head against the oracle chain, every output within the fp32 bar."""
    from helpers import head_cfg, oracle_head, tie_aware_topk_match
    from oracle.backbone import OracleResNet50, seeded_backbone_state
    from pairnet_amd import build_detector, pairnet_r50
    head_o, sd, _ = oracle_head(1234)
    bsd = seeded_backbone_state(31)
    bb_o = OracleResNet50()
    bb_o.load_state_dict(bsd)
    det = build_detector(pairnet_r50())
    det.backbone.load_state_dict(bsd)
    det.bbox_head.load_state_dict(sd)
    det.to(DEV)
    img = seeded.uniform(np.random.default_rng(H * 1000 + W), (1, 3, H, W), -2.0, 2.0)
    metas = [dict(img_shape=(H, W, 3), scale_factor=[1.0] * 4)]
    with torch.no_grad():
        feats_o = [f.contiguous() for f in bb_o(img)]
    trace = {}
    ref_cls, ref_masks = head_o.forward(feats_o, metas, trace=trace)
    feats = det.extract_feat(img.to(DEV))
    for f, o in zip(feats, feats_o):
        assert tuple(f.shape) == tuple(o.shape)
        assert _err(f, o) < 1e-4 * float(o.abs().max())
    res = det.simple_test(img.to(DEV), metas)
    pl = det.bbox_head._last_plan
    assert pl.wino == (pl.hw2[0] % 2 == 0 and pl.hw2[1] % 2 == 0)
    for k, got in (("cls", pl.cls), ("importance", pl.imp)):
        assert _err(got, ref_cls[k]) < 1e-3, k
    scale = max(1.0, float(ref_masks["mask"].abs().max()))
    assert _err(pl.MP.view_as(ref_masks["mask"]), ref_masks["mask"]) < 1e-3 * scale
    ok, exact = tie_aware_topk_match(ref_cls["importance"][0].numpy(), trace["topk_idx"][0].numpy(),
                                     pl.topk_idx[0].cpu().numpy(), 3e-6)
    assert ok
    if exact == 100:
        assert _err(pl.rel, ref_cls["rel"]) < 1e-3
    assert res[0].masks.shape == (200, H, W) and res[0].pan_results.shape == (H, W)


def test_many_shapes_share_one_arena_per_slot():
    """VERDICT r4 next 1 (was ADVICE r2): a keep-ratio evaluation pass meets hundreds of
    distinct shapes (tools/test.py:199-267 with Pad(size_divisor=1)).  Plans are VIEWS of one
    arena per slot, sized for the envelope of the shapes met: after `reserve` (or after the
    largest shapes have passed) a new shape allocates nothing, waits for nothing and evicts
    nothing; device memory is independent of the number of shapes; every image's result is
    bitwise the one a fresh detector gives for that shape alone; graphs are captured on a
    shape's second sight -- this caller is single-stream, i.e. always at a quiet point
    (plans.quiet) -- and replayed from then on (zero recaptures)."""
    from oracle.backbone import seeded_backbone_state
    from pairnet_amd import build_detector, pairnet_r50

    def make():
        det = build_detector(pairnet_r50())
        det.backbone.load_state_dict(seeded_backbone_state(41))
        det.to(DEV)
        return det
    det = make()
    head, net = det.bbox_head, det.backbone
    head.use_graphs = net.use_graphs = True
    shapes = [(96 + 8 * (i % 7), 128 + 16 * (i % 5)) for i in range(35)]   # 35 distinct (H, W)
    assert len(set(shapes)) == 35
    env = [(max(h for h, _ in shapes), max(w for _, w in shapes))]
    reserved = det.reserve(env, orig_sizes=env)
    assert reserved == net.arena_bytes() + head.arena_bytes() > 0
    grows = [a.grows for a in list(head._arenas.values()) + list(head._post_arenas.values())
             + list(net._arenas.values())]

    def run(d, H, W, seed, reps):
        img = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(seed)).to(DEV)
        metas = [dict(img_shape=(H, W, 3), scale_factor=[1.0] * 4)]
        out = None
        for _ in range(reps):                   # eager, capture, replay
            r = d.bbox_head.simple_test_bboxes(d.extract_feat(img), metas)[0]
            out = [t.clone() for t in (r[1], r[7], r[4])]
        return out
    torch.cuda.synchronize()
    from pairnet_amd.head import CrossHead2
    caps = CrossHead2.captures
    syncs = []
    real_sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: (syncs.append(1), real_sync(*a, **k))[1]
    try:
        outs, mem = [], []
        for i, (H, W) in enumerate(shapes):
            outs.append(run(det, H, W, 1 + i, 3))
            mem.append(torch.cuda.memory_allocated())
    finally:
        torch.cuda.synchronize = real_sync
    # the only device-wide waits are the ones in front of a graph capture (a single-stream
    # caller is always at a quiet point): none for plans, none for new shapes as such
    assert len(syncs) == CrossHead2.captures - caps >= 35 * 3
    assert [a.grows for a in list(head._arenas.values()) + list(head._post_arenas.values())
            + list(net._arenas.values())] == grows     # the reserved arenas never grew
    assert head._plans.evictions == 0 and net._plans.evictions == 0
    assert all(pl.graph_a is not None and pl.graph_b is not None for pl in head._plans.values())
    assert getattr(head, "recaptures", 0) == 0
    # memory: the arenas + the shared position tables (one set per shape, LRU-bounded) +
    # the clones of this loop; independent of how many shapes have passed
    # (e[3]: the encoder table once more in the order gemm_s3's `out + pos` epilogue reads)
    pe = max(sum(t.numel() * 4 for t in [e[0]] + e[1] + [e[3]]) for e in head._pe.values())
    assert len(head._pe) <= head.PE_SHAPES
    assert max(mem[12:]) - mem[11] <= head.PE_SHAPES * pe + (1 << 20), (mem[11], max(mem[12:]))
    # the first shape again: its plan is still there, replayed, same bits
    again = run(det, *shapes[0], 1, 1)
    assert all(torch.equal(a, b) for a, b in zip(outs[0], again))
    # against a fresh detector that only ever sees that one shape (eager): no aliasing error
    for i in (0, 17, 34):
        fresh = make()
        want = run(fresh, *shapes[i], 1 + i, 1)
        assert all(torch.equal(a, b) for a, b in zip(outs[i], want)), i
    # a shape beyond the envelope: the arenas grow once (a device wait), results stay right
    big = (env[0][0] + 32, env[0][1] + 32)
    got = run(det, *big, 99, 3)
    assert head._arenas[0].grows == grows[0] + 1
    want = run(make(), *big, 99, 1)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    again = run(det, *shapes[5], 6, 2)                 # (plans of the slot were rebuilt)
    assert all(torch.equal(a, b) for a, b in zip(outs[5], again))


def test_product_loop_takes_decoded_images_of_mixed_sizes():
    """`dist.multi_gpu_test` fed like the reference's loader feeds `multi_gpu_test`
    (tools/test.py:199-267): DECODED uint8 images of different original sizes; the detector's
    own test pipeline (Resize keep-ratio -> Normalize -> Pad -> collate, one kernel per image on
    the batch's stage-A stream, one grow-only buffer per stream) runs in front of the backbone.
    Every record equals the synchronous `test_pipeline -> simple_test` path, one image per step
    and two per step (unequal sizes zero-padded to the batch maximum, like mmcv's collate)."""
    from oracle.backbone import seeded_backbone_state
    from pairnet_amd import TestPipeline, build_detector, pairnet_r50
    from pairnet_amd.dist import multi_gpu_test, pack_triplets
    det = build_detector(pairnet_r50())
    det.backbone.load_state_dict(seeded_backbone_state(41))
    det.bbox_head.init_weights(seed=3)
    det.to(DEV)
    det.test_pipeline = TestPipeline(img_scale=(224, 160), device=DEV)
    g = torch.Generator().manual_seed(9)
    sizes = [(60, 80), (80, 60), (64, 64), (60, 80), (50, 90), (75, 100), (90, 50)]
    images = [torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8) for h, w in sizes]
    images[1] = images[1].numpy()                        # (numpy arrays and host tensors both work)
    head = det.bbox_head

    def want(batch):
        img, metas = det.test_pipeline.batch(batch, slot=9)
        res = head.simple_test(det.extract_feat(img), metas)
        sub, obj = head.pair_positions()
        return [pack_triplets(r[1].cpu(), r[7].cpu(), sub[j].cpu(), obj[j].cpu())
                for j, r in enumerate(res)], metas
    single = [want([im])[0][0] for im in images]
    _, metas = want([images[0]])
    assert metas[0]["ori_shape"] == (60, 80, 3) and metas[0]["img_shape"] == (160, 213, 3)
    data = [(im, None) for im in images]
    out = multi_gpu_test(det, data, depth=3, calibrate=False)
    assert out["records"].shape[0] == len(images)
    for i, w in enumerate(single):
        assert torch.equal(out["records"][i].cpu(), w), i
    out2 = multi_gpu_test(det, data, depth=3, calibrate=False, samples_per_gpu=2)
    for grp in ((0, 1), (2, 3), (4, 5), (6,)):
        ws, ms = want([images[i] for i in grp])
        assert len({m["batch_input_shape"] for m in ms}) == 1          # one padded batch tensor
        for j, i in enumerate(grp):
            assert torch.equal(out2["records"][i].cpu(), ws[j]), i
    assert len(det.test_pipeline._slots) <= 3                # one buffer per stage-A stream (+ slot 9)


def test_image_sizes_sharing_a_feature_pyramid_keep_their_own_stage_graphs():
    """Image widths 129 and 130 give the same feature pyramid -- one head plan -- but two
    backbone plans, i.e. two sets of feature buffers.  The head keeps a stage-A graph per
    buffer set (by pointer), so alternating the two sizes replays both instead of dropping
    and re-capturing one graph every time (20 recaptures in the first round-5 shape-mix run:
    800 x 1201 and 800 x 1202 share a pyramid)."""
    from oracle.backbone import seeded_backbone_state
    from pairnet_amd import build_detector, pairnet_r50

    def make():
        det = build_detector(pairnet_r50())
        det.backbone.load_state_dict(seeded_backbone_state(41))
        det.to(DEV)
        return det
    det = make()
    head, net = det.bbox_head, det.backbone
    head.use_graphs = net.use_graphs = True
    sizes = [(96, 129), (96, 130)]
    assert net.feature_shapes(*sizes[0]) == net.feature_shapes(*sizes[1])
    det.reserve([(96, 130)], orig_sizes=[(96, 130)])     # (a growing arena moves the buffers)
    imgs = [torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(i)).to(DEV)
            for i, (H, W) in enumerate(sizes)]
    metas = [[dict(img_shape=(H, W, 3), scale_factor=[1.0] * 4)] for H, W in sizes]
    outs = [[], []]
    for rep in range(4):
        for i in (0, 1):
            r = head.simple_test_bboxes(det.extract_feat(imgs[i]), metas[i])[0]
            outs[i].append([t.clone() for t in (r[1], r[7], r[4])])
    assert len(head._plans) == 1 and len(net._plans) == 2
    pl = head._last_plan
    assert len(pl.graphs_a) == 2 and all(e["graph"] is not None for e in pl.graphs_a.values())
    assert getattr(head, "recaptures", 0) == 0
    for i in (0, 1):
        fresh = make()
        r = fresh.bbox_head.simple_test_bboxes(fresh.extract_feat(imgs[i]), metas[i])[0]
        for got in outs[i]:
            assert all(torch.equal(a, b) for a, b in zip(got, (r[1], r[7], r[4]))), i


def test_plan_cache_eviction_parks_busy_plans():
    """More live (shape, slot) plans than the cache holds: the oldest is evicted without a
    host wait -- parked until the streams it ran on have passed -- and a shape that comes
    back is served again (views + a new capture), bit for bit."""
    from oracle.backbone import seeded_backbone_state
    from pairnet_amd import build_detector, pairnet_r50
    det = build_detector(pairnet_r50())
    det.backbone.load_state_dict(seeded_backbone_state(41))
    det.to(DEV)
    head, net = det.bbox_head, det.backbone
    head.use_graphs = net.use_graphs = True
    head._plans.max_plans = net._plans.max_plans = 3
    shapes = [(96 + 8 * i, 128 + 16 * (i % 3)) for i in range(8)]
    det.reserve([(160, 160)], orig_sizes=[(160, 160)])    # (arena growth also drops plans)
    first = {}
    for rep in range(2):
        for i, (H, W) in enumerate(shapes):
            img = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(i)).to(DEV)
            metas = [dict(img_shape=(H, W, 3), scale_factor=[1.0] * 4)]
            for _ in range(2):
                r = head.simple_test_bboxes(det.extract_feat(img), metas)[0]
            out = [t.clone() for t in (r[1], r[7], r[4])]
            if rep == 0:
                first[i] = out
            else:
                assert all(torch.equal(a, b) for a, b in zip(first[i], out)), i
    assert len(head._plans) <= 3 and head._plans.evictions == 13 and net._plans.evictions == 13
    assert head._arenas[0].grows == 1 and net._arenas[0].grows == 1
    torch.cuda.synchronize()
    assert head._plans.reap() == 0 and net._plans.reap() == 0


def test_simple_test_mask_arrays_are_private_and_recycled():
    """`PSGTr.simple_test` fetches the masks bit-packed into arrays from a pool that an array
    only returns to when it is garbage collected: results the caller still holds are never
    overwritten, dropped ones are reused, and the values are those of `.cpu().numpy()`."""
    import gc
    from oracle.backbone import seeded_backbone_state
    from pairnet_amd import build_detector, pairnet_r50
    det = build_detector(pairnet_r50())
    det.backbone.load_state_dict(seeded_backbone_state(41))
    det.bbox_head.init_weights(seed=3)
    det.to(DEV)
    H, W = 416, 544            # 200 x 208 x 272 masks: above the packed-transfer threshold
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0] * 4)]
    g = torch.Generator().manual_seed(5)
    imgs = [torch.randn(1, 3, H, W, generator=g).to(DEV) for _ in range(3)]
    want = []
    for im in imgs:
        tup = det.bbox_head.simple_test(det.extract_feat(im), metas, rescale=False)
        want.append(tup[0][3].cpu().numpy().copy())
    assert want[0].size >= 1 << 20 and not np.array_equal(want[0], want[1])
    assert 0.02 < want[0].mean() < 0.98                             # (not a constant mask)
    held = [det.simple_test(im, metas)[0] for im in imgs]          # all three alive
    for r, w in zip(held, want):
        assert r.masks.dtype == np.bool_ and np.array_equal(r.masks, w)
    ptrs = {r.masks.ctypes.data for r in held}
    assert len(ptrs) == 3                                           # three private arrays
    view = held[0].masks[5]                                         # a view keeps its array alive
    p0 = held[0].masks.ctypes.data
    del held
    gc.collect()
    again = [det.simple_test(im, metas)[0] for im in imgs]
    assert np.array_equal(view, want[0][5])                         # ... and was not overwritten
    assert p0 not in {r.masks.ctypes.data for r in again}
    assert {r.masks.ctypes.data for r in again} & ptrs              # dropped arrays came back
    for r, w in zip(again, want):
        assert np.array_equal(r.masks, w)


@pytest.mark.parametrize("copy", [True, False])
def test_detector_stream_equals_simple_test(copy):
    """`PSGTr.stream` (pipeline + ResultStreamer behind one generator) yields, in order, what
    `simple_test` returns image by image; with copy=True the arrays outlive the ring."""
    from oracle.backbone import seeded_backbone_state
    from pairnet_amd import build_detector, pairnet_r50
    det = build_detector(pairnet_r50())
    det.backbone.load_state_dict(seeded_backbone_state(41))
    det.bbox_head.init_weights(seed=3)
    det.to(DEV)
    H, W = 160, 224
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0] * 4)]
    g = torch.Generator().manual_seed(9)
    imgs = [torch.randn(1, 3, H, W, generator=g).to(DEV) for _ in range(9)]
    fields = ("refine_bboxes", "labels", "rel_pair_idxes", "rel_dists", "rel_labels",
              "pan_results", "masks")
    want = [{k: np.array(getattr(det.simple_test(im, metas, rescale=True)[0], k)) for k in fields}
            for im in imgs]
    got = []
    for results in det.stream(((im, metas) for im in imgs), rescale=True, ring=4, copy=copy):
        assert len(results) == 1
        got.append(results[0] if copy else
                   {k: np.array(getattr(results[0], k)) for k in fields})
    assert len(got) == len(want)
    for r, w in zip(got, want):
        for k in fields:
            v = np.array(getattr(r, k)) if copy else r[k]
            assert v.dtype == w[k].dtype and np.array_equal(v, w[k]), k
    # the detector is left as it was found
    assert det.bbox_head.use_graphs is False and det.bbox_head.grid_reserve == 0
    assert det.backbone.grid_reserve == 0
