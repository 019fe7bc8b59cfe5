"""Headline benchmark: images/sec of the Pair-Net inference path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one batch through `PSGTr.simple_test`'s device work (psgtr.py:148-156):
an 800x1333 image tensor already resident in HBM -> native ResNet-50 -> `CrossHead2.
simple_test_bboxes` (pairnet_head.py:926-930: pixel decoder, 9-layer masked decoder,
PPN / Matrix Learner / top-k, 6-layer relation decoder, get_bboxes), 100 object / 100
relation queries, fp32 -- BASELINE.json configs[1] on each GPU; consecutive steps take
DIFFERENT images (a rotating pool of `--pool` distinct tensors, so that neither weights nor
activations of "the" image can sit in the 256 MB MALL); attention masks are computed in the
reference's operation order (`--mask-order`).  The secondary leg
`simple_test_incl_result_d2h` adds the rest of `simple_test`: the panoptic-loop status check
and `triplet2Result` of every field to the host (psgtr.py:15-51; pinned ring buffers, the
bool masks as bits over PCIe, `Result` objects built per image); further legs: the opt-in
mask shortcut, two images per GPU, attention vs mask density, deformable sampling vs offset
spread.  `--path head` times the
head alone on a resident feature pyramid (round 1's headline; reported by the default run
as the secondary `head_only`).  For N > 1 one rank per GPU over RCCL: started by
torch.distributed.run, or by this script itself when WORLD_SIZE is unset (plain `python
bench.py --gpus 8` re-executes through torch.distributed.run); every rank processes its
own images (weak scaling, no data-path collective) and the predicted triplet records are
all-gathered once per step.

Rank 0 prints ONE JSON line: the contract fields plus
  roofline      the dominant kernel's achieved rate, from HIP events recorded around
                its launches on the launching stream (eager steps right after the timed
                region)
  cpu_baseline  the CPU oracle (oracle/backbone.py + oracle/head.py, kind "port") on the
                host cores, same weights and inputs, bounded sample (N == 1 only)
"""
import argparse
import gc
import math
import glob
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (no sparsity)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec
# kernels whose matrix work is ISSUED as six bf16 MFMA products per fp32 multiply-add
# (csrc/gemm_s3.hip); hip.py's flop count for them is the fp32-equivalent 2 M N K
BF16X3_KERNELS = ("k_gemm_s3", "k_gemm_s3<ln>")
GEMM_ARITHMETIC = {
    "bf16x3": "pixel-decoder encoder GEMMs (and a Swin backbone's block GEMMs): fp32 = 3 x bf16 "
              "exact split, 6 products, fp32 accumulate (csrc/gemm_s3.hip); everything else: "
              "exact-fp32 MFMA",
    "fp32": "exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere"}
HBM_KERNELS = ("k_msda",)


def feature_shapes(h, w):
    h, w = (h + 1) // 2, (w + 1) // 2
    h, w = (h + 1) // 2, (w + 1) // 2
    out = []
    for _ in range(4):
        out.append((h, w))
        h, w = (h + 1) // 2, (w + 1) // 2
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-execute as N ranks, one per GPU
    (what tools/dist_test.sh:7 does for the reference's test script)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def latest_pmc():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    return files[-1] if files else None


def gather_roof(dev, hip, lines_log2=(12, 14, 18), reps=20):
    """In-run micro-probe (csrc/msda.hip k_gather_probe): the rate at which THIS board delivers
    the deformable-sampling access pattern -- 8 lanes x 16 B per random 128-byte line, 12 lines
    in flight per lane group, nothing else -- for a window of 2^k lines: 512 KB (L2-resident
    everywhere), 2 MB (about one XCD band of the value map: the kernel's working set per L2)
    and 32 MB (HBM).  Returns {window_kb: TB/s}; one layer moves 21 950 x 8 x 12 x 4 lines."""
    nwg = 21950 * 4
    lines = torch.zeros((1 << max(lines_log2)) * 32, device=dev)
    idx = torch.randint(0, 1 << 30, (nwg * 32 * 12,), device=dev, dtype=torch.int32)
    out = torch.empty(nwg * 256, device=dev)
    res = {}
    for k in lines_log2:
        mask = (1 << k) - 1
        for _ in range(3):
            hip.gather_probe(lines, idx, out, nwg, mask)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            hip.gather_probe(lines, idx, out, nwg, mask)
        e1.record()
        e1.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3 / reps
        res[(1 << k) * 128 // 1024] = nwg * 256 * 12 * 16 / sec / 1e12
    return res


def bench_bbox(args, dev, rank, world):
    """Secondary measurement: the box-trunk sibling
    configs/deformable_detr/cross_r101_vg.py -- image tensor -> ResNet-101 (C3-C5) ->
    ChannelMapper -> CrossHeadBBox (two-stage Deformable-DETR trunk, PPN, relation decoder) ->
    get_bboxes, scheduled like the headline path (hipGraph replay per stage under
    `PipelinedHead`; `--no-pipeline`: one stream, one image in flight).  Not the headline (BASELINE.json names the Mask2Former path); same JSON contract."""
    import torch.distributed as dist
    from pairnet_amd import build_detector, cross_r101_vg, hip
    det = build_detector(cross_r101_vg().model).to(dev)
    det.bbox_head.init_weights(seed=0)
    det.bbox_head.to(dev)
    det.bbox_head.use_graphs = det.backbone.use_graphs = not args.no_graphs
    B, H, W = args.batch, args.height, args.width
    img = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(1000 + rank)).to(dev)
    metas = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=[2.083] * 4)] * B

    from pairnet_amd import PipelinedHead
    head, depth = det.bbox_head, args.depth
    eng = None if args.no_pipeline else PipelinedHead(
        head, depth=depth, a_streams=args.a_streams,
        **({} if args.grid_trim is None else dict(grid_trim=args.grid_trim)))
    if eng is not None:     # the producers in front of stage A run on the stage-A streams too
        det.backbone.grid_reserve = det.neck.grid_reserve = eng.grid_reserve
    count = [0]

    def step():
        """One image: backbone + neck + stage A on this batch's stage-A stream (per-slot
        buffers), the query chain of older batches on the chain streams (pipeline.py)."""
        if eng is None:
            return head.simple_test_bboxes(det.extract_feat(img), metas, rescale=True)
        i = count[0]
        count[0] += 1
        with torch.cuda.stream(eng.streams_a[i % len(eng.streams_a)]):
            x = det.backbone(img, slot=i % depth)
            feats = det.neck(tuple(x[j] for j in det.out_indices), slot=i % depth)
            return eng.submit(feats, metas, rescale=True)

    def drain():
        if eng is None:
            return []
        with torch.cuda.stream(eng.streams_a[0]):
            return eng.flush()
    for _ in range(2 * depth if eng is not None else 0):   # graphs are captured at quiet points
        step()                      # only (LABNOTES R5.9): one image at a time through every slot
        torch.cuda.synchronize()
        drain()
        torch.cuda.synchronize()
    if eng is not None:
        # the stream -> hardware-queue placement is chosen empirically, as for the headline
        # path (pipeline.py: the order in which streams are first used decides HIP's mapping --
        # the one-at-a-time warm-up above alone cost 12 %: 160 instead of 182 images/s)
        eng.calibrate(None, metas, submit=step)
    for _ in range(max(args.warmup, depth)):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    tail = drain()
    res = tail[-1] if tail else res
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    nccl = os.environ.get("PAIRNET_DIST_BACKEND", "nccl") == "nccl"
    dt = torch.tensor([time.perf_counter() - t0], device=dev if nccl else "cpu",
                      dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt)
    # scheduling only: the last pipelined batch against a plain single-stream call
    labels, dists = res[0][1].clone(), res[0][5].clone()
    head.use_graphs = det.backbone.use_graphs = False
    plain = head.simple_test_bboxes(det.extract_feat(img), metas, rescale=True)
    torch.cuda.synchronize()
    check = bool(torch.equal(labels, plain[0][1]) and torch.equal(dists, plain[0][5]))
    eng = None
    # per-kernel times of one eager step (HIP events around every launch)
    step()
    hip.TIMER = hip.KernelTimer()
    for _ in range(3):
        step()
    agg = hip.TIMER.summary()
    hip.TIMER = None
    prof = {k: {"ms_per_step": v["ms"] / 3, "launches_per_step": v["launches"] // 3,
                "tflops": v["flops"] / v["ms"] * 1e-9 if v["ms"] else 0.0,
                "gbs": v["bytes"] / v["ms"] * 1e-6 if v["ms"] else 0.0}
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
    out = {"metric": "images/sec (whole node), CrossHeadBBox 300-proposal 800x1333, MI355X",
           "value": world * B * args.steps / dt, "unit": "images/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": "cross_r101_vg: ResNet-101 (C3-C5) -> ChannelMapper -> CrossHeadBBox "
                                  "(6+6-layer two-stage box-refining Deformable-DETR trunk, 300 "
                                  "proposals -> 100 kept queries -> PPN / Matrix Learner / top-k -> "
                                  "6-layer relation decoder -> get_bboxes), image tensor -> triplets, "
                                  "bs=%d per GPU, %dx%d, random-init weights" % (B, H, W),
                      "head": "bbox", "global_batch": world * B, "image": [H, W],
                      "parallelism": "dp%d" % world,
                      "schedule": "one stream, backbone and head as hipGraph replays" if args.no_pipeline
                      else "hipGraph replay per stage, %d-stream pipeline: backbone + neck + stage A "
                           "(encoder, proposals) of consecutive images alternate between %d streams, "
                           "their query chains run on %d more" % (depth, args.a_streams,
                                                                  depth - args.a_streams)},
           "pipeline_check": "labels / rel_dists of the last pipelined image are bitwise the eager "
                             "single-stream result" if check else "MISMATCH",
           "kernel_profile": prof}
    if not check:
        raise SystemExit("pipelined result differs from the eager one")
    import torch.distributed as dist
    if world > 1 and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    try:      # (the JSON line last on stdout: see main())
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # pragma: no cover
        pass
    if rank == 0:
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--path", choices=["image", "head"], default="image",
                    help="image: image tensor -> backbone -> head -> triplets (headline); "
                         "head: the head alone on a resident feature pyramid")
    ap.add_argument("--pool", type=int, default=8,
                    help="distinct input images the steps rotate over")
    ap.add_argument("--head", choices=["pairnet", "baseline", "psgtr2", "bbox"], default="pairnet",
                    help="pairnet = CrossHead2 (the headline); baseline / psgtr2 = the sibling "
                         "heads CrossHeadBaseline / PSGTrHead2 on the same trunk (not the "
                         "headline metric)")
    ap.add_argument("--queries", type=int, default=100, help="object queries (BASELINE configs[3]: 200)")
    ap.add_argument("--in-channels", default="256,512,1024,2048",
                    help="backbone channel widths (Swin-L: 192,384,768,1536)")
    ap.add_argument("--conv", choices=["winograd", "winograd4", "direct"], default=None,
                    help="algorithm of the 3x3 FPN convolution (default: the head's)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run the stages of consecutive batches back to back on one stream")
    ap.add_argument("--depth", type=int, default=4, help="batches in flight in the pipeline")
    ap.add_argument("--a-streams", type=int, default=2,
                    help="streams that backbone + stage A of consecutive batches alternate between")
    ap.add_argument("--no-graphs", action="store_true", help="launch every kernel eagerly")
    ap.add_argument("--mask-order", choices=["reference", "dense", "resampled"],
                    default="reference",
                    help="attention masks: 'reference' = the reference's operation order "
                         "(full-size mask logits -> bilinear resize -> threshold), evaluated only "
                         "at the logits the resize reads (default); 'dense' = the same with all "
                         "full-size logits (bit-identical); 'resampled' = the opt-in shortcut: "
                         "logits against the once-resampled mask feature")
    ap.add_argument("--rccl-one-rank", action="store_true",
                    help="with --gpus 1: initialise RCCL with world size 1 and run the per-step "
                         "all-gather, the barriers and the timing reduction through it")
    ap.add_argument("--grid-trim", type=int, default=None,
                    help="persistent-GEMM workgroup slots left free for the query chains")
    ap.add_argument("--enc-fused-ln", default=None,
                    help="encoder sites whose Linear + residual + LayerNorm run as one launch: "
                         "'proj,ffn' (default), 'proj', 'ffn' or 'none' (A/B of csrc/gemm_ln.hip)")
    ap.add_argument("--set", action="append", default=[], metavar="ATTR=VALUE",
                    help="A/B aid: set a boolean / integer scheduling attribute of the head before "
                         "the run (e.g. --set group_input_convs=0)")
    ap.add_argument("--bb-set", action="append", default=[], metavar="ATTR=VALUE",
                    help="integer attribute of the native ResNet backbone (tuning A/B)")
    ap.add_argument("--mix-images", type=int, default=240,
                    help="images of the shape_mix_product_loop leg (>= 200 by default)")
    ap.add_argument("--gemm-arithmetic", choices=["bf16x3", "fp32"], default="bf16x3",
                    help="encoder GEMMs: bf16x3 = fp32 operands as three exact bf16 planes, six "
                         "bf16 MFMA products, fp32 accumulation (default); fp32 = the exact-fp32 "
                         "MFMA kernels of rounds 1-5")
    ap.add_argument("--no-sub-legs", action="store_true",
                    help="skip the Swin-L / 200-query and box-trunk legs (each a short run of this "
                         "script in a child process)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    # one rank per GPU; (for a functional check of the multi-rank control flow on a
    # single-GPU box: PAIRNET_DIST_BACKEND=gloo lets several ranks share cuda:0)
    backend = os.environ.get("PAIRNET_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit("--gpus %d but only %d GPU(s) visible" % (world, ndev))
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    numa_bound = None
    if world > 1 or os.environ.get("PAIRNET_NUMA_BIND"):
        from pairnet_amd.dist import bind_to_gpu_numa
        numa_bound = bind_to_gpu_numa(dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = dict(device_id=dev) if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    # --rccl-one-rank: the multi-GPU code path on ONE GPU -- RCCL communicator (and its
    # watchdog thread) alive during graph capture and replay, the triplet records of every step
    # through all_gather_into_tensor on the side stream, barrier + MAX reduction around the
    # timed region -- with world size 1 (what a single-GPU box can execute of it)
    one_rank = bool(args.rccl_one_rank) and world == 1 and backend == "nccl"
    if one_rank:
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port_ = s_.getsockname()[1]
        s_.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port_)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    if args.head == "bbox":
        return bench_bbox(args, dev, rank, world)
    from pairnet_amd import (CrossHead2, CrossHeadBaseline, PSGTrHead2, PipelinedHead,
                             ResNet50Hip, SwinTransformerHip, baseline_head_cfg, hip,
                             pairnet_head_cfg, psgtr2_head_cfg, swin_backbone_cfg)
    from pairnet_amd.dist import TripletBatch, TripletCollector

    sibling = args.head != "pairnet"
    chans = tuple(int(c) for c in args.in_channels.split(","))
    cfg = {"pairnet": pairnet_head_cfg, "baseline": baseline_head_cfg,
           "psgtr2": psgtr2_head_cfg}[args.head](in_channels=chans, num_obj_query=args.queries)
    if args.head == "baseline":
        cfg["num_rel_query"] = args.queries
    cfg.pop("type")
    head = {"pairnet": CrossHead2, "baseline": CrossHeadBaseline,
            "psgtr2": PSGTrHead2}[args.head](**cfg)
    ident = torch.arange(head.num_obj_query, device=dev).unsqueeze(0).expand(args.batch, -1)
    pair_ids = head.pair_positions      # the query rows of each image's R triplets
    head.init_weights(seed=0)
    head.to(dev)
    MASK_ORDER = {"reference": True, "dense": "full", "resampled": False}
    head.exact_mask_order = MASK_ORDER[args.mask_order]
    head.gemm_arithmetic = args.gemm_arithmetic
    if args.conv:
        head.conv_algo = args.conv
    for kv in args.set:
        k, v = kv.split("=", 1)
        if not hasattr(head, k):
            raise SystemExit("--set: the head has no attribute %r" % k)
        setattr(head, k, type(getattr(head, k))(int(v)))
    if args.enc_fused_ln is not None:
        head.enc_fused_ln = tuple(t for t in args.enc_fused_ln.split(",") if t in ("proj", "ffn"))
    head.use_graphs = not args.no_graphs
    engine = None if args.no_pipeline else PipelinedHead(
        head, depth=args.depth, a_streams=args.a_streams,
        **({} if args.grid_trim is None else dict(grid_trim=args.grid_trim)))
    pool = []
    B, H, W = args.batch, args.height, args.width
    g = torch.Generator().manual_seed(1000 + rank)
    sf = 2.083
    metas = [dict(img_shape=(H, W, 3), scale_factor=[sf, sf, sf, sf])] * B
    R = head.num_rel_query

    # the backbone whose channel widths the head was built for: ResNet-50 (pairnet.py)
    # or Swin-T/B/L (pairnet_swinb.py; BASELINE configs[3] is Swin-L), random weights
    swin = {96: "T", 128: "B", 192: "L"}.get(chans[0])
    backbone, bname, img_cpu, img = None, None, None, None
    if args.path == "image":
        if swin:
            scfg = swin_backbone_cfg(swin)
            scfg.pop("type")
            backbone, bname = SwinTransformerHip(**scfg).to(dev), "Swin-%s" % swin
            backbone.gemm_arithmetic = args.gemm_arithmetic     # (its block GEMMs: same switch)
        else:
            backbone, bname = ResNet50Hip().to(dev), "ResNet-50"
            for kv in args.bb_set:
                k, v = kv.split("=", 1)
                if not hasattr(backbone, k):
                    raise SystemExit("--bb-set: the backbone has no attribute %r" % k)
                setattr(backbone, k, int(v))
            backbone.use_graphs = not args.no_graphs
        if engine is not None:   # the backbone runs on the stage-A streams: same free slots
            backbone.grid_reserve = engine.grid_reserve
        # a pool of DISTINCT normalised image batches; step i takes pool[i % len(pool)]
        pool_cpu = [torch.randn(B, 3, H, W, generator=g) for _ in range(max(1, args.pool))]
        pool = [t.to(dev) for t in pool_cpu]
        img_cpu, img = pool_cpu[0], pool[0]
        feats_cpu = None
        # (own copies: the backbone's outputs are views of per-slot buffers it reuses)
        feats = [f.clone(memory_format=torch.preserve_format) for f in backbone(img)]
    else:
        feats_cpu = [torch.relu(torch.randn(B, c, h, w, generator=g))
                     for c, (h, w) in zip(chans, feature_shapes(H, W))]
        feats = [f.to(dev) for f in feats_cpu]

    # The multi-GPU leg is the product's own: `dist.TripletCollector` (what
    # `dist.multi_gpu_test` is built on) packs each batch's triplet records on the stream that
    # produced them -- the chain stream of a pipelined result -- and all-gathers the records of
    # the batch `depth` steps back on a side stream: that collective's inputs have long been
    # written, so it never makes a hardware queue wait (a side-stream command that waits for
    # the NEWEST chain blocks the pipeline stream HIP maps onto the same queue: 188 instead of
    # 203 images/s per GPU), and neither the collective nor the other ranks' arrival at it
    # stalls a compute stream.
    collector = TripletCollector(head, depth=args.depth, n_local=B, force_collective=one_rank,
                                 host_staging=backend != "nccl") if world > 1 or one_rank else None
    gatherer = collector.gatherer if collector is not None else None

    def gather(res, sub_pos, obj_pos):
        if collector is not None and res is not None:
            src = getattr(res, "pipeline_stream", None) if engine is not None else None
            collector.add(TripletBatch(
                res, sub_pos, obj_pos, stream=src,
                release=(lambda s, r=res: engine.consumed(r, s)) if engine is not None else None))

    def gather_flush():
        if collector is not None:
            collector.finish()

    def step(with_backbone=args.path == "image"):
        """One batch.  Pipelined: backbone + stage A of this batch are queued on the stage-A
        stream beside the query chains of the two previous batches (results arrive two
        steps late; drain() completes the batches still in flight)."""
        im = pool[nstep[0] % len(pool)] if with_backbone else None
        nstep[0] += 1
        if engine is None:
            res = head.simple_test_bboxes(backbone(im) if with_backbone else feats, metas)
        elif with_backbone:
            # two chip-filling kernel sequences on different streams time-slice badly
            # (LABNOTES.md 6a): the backbone goes in front of stage A on its stream
            # (each stage-A stream has its own set of backbone buffers)
            sl = engine.count % len(engine.streams_a)
            with torch.cuda.stream(engine.streams_a[sl]):
                res = engine.submit(backbone(im, slot=sl), metas)
                if res is not None:   # (results are ordered behind the submitting stream)
                    gather(res, *pair_ids(head._last_plan))
                    if on_result[0] is not None:
                        on_result[0](res)
            return res
        else:
            res = engine.submit(feats, metas)
        if res is not None:
            gather(res, *pair_ids(head._last_plan))
            if on_result[0] is not None:
                on_result[0](res)
        return res

    nstep = [0]            # steps issued so far (selects the pool image)
    on_result = [None]     # optional consumer of every result list (legs below)

    def drain():
        if engine is not None:
            sa0 = engine.streams_a[0] if args.path == "image" else torch.cuda.current_stream()
            with torch.cuda.stream(sa0):
                while engine.queue:
                    res = engine._finish(engine.queue.pop(0))
                    gather(res, *pair_ids(head._last_plan))
                    if on_result[0] is not None:
                        on_result[0](res)
        gather_flush()             # the records still in the gatherer's ring

    # ---- warm-up (graph capture happens in the first two steps), then the stream ->
    # hardware-queue placement of the pipeline is chosen empirically (pipeline.py) ----
    # (setup, not warm-up: buffers are planned on the first call of a shape and the hipGraphs
    # are captured on the second, per pipeline slot -- two passes over the slots)
    # (graphs are captured at QUIET points only -- no other stream executing, LABNOTES R5.9 --
    # so the set-up runs one batch at a time: stage A, device wait, chain + get_bboxes, wait)
    for _ in range(2 * args.depth if engine is not None else 2):
        step()
        torch.cuda.synchronize()
        drain()
        torch.cuda.synchronize()
    calibration = None
    if engine is not None:
        calibration = engine.calibrate(feats, metas, submit=step)
    # Python's cyclic GC (gen-2 passes of 50-100 ms over torch's object graph) would
    # land inside the timed region at random: collect now, then keep it off, as a
    # serving loop would.
    gc.collect()
    gc.freeze()
    gc.disable()
    # ---- the W untimed warm-up steps, directly in front of the timed region: the setup
    # above leaves the GPU idle for tens of ms (host-side calibration bookkeeping, the GC
    # pass), and the first 20 steps after >= 50 ms of idle take 3 % longer than the same 20
    # steps issued back to back (tools/filldrain_probe.py: 101.1 vs 97.9 ms) ----
    for _ in range(args.warmup):
        step()
    drain()

    host_submit = [0.0, 0]           # host seconds spent inside step() / steps (this rank)

    def timed(n, **kw):
        torch.cuda.synchronize()
        if world > 1 or one_rank:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            step(**kw)
        host_submit[0] += time.perf_counter() - t0
        host_submit[1] += n
        drain()                      # the last batch finishes inside the timed region
        torch.cuda.synchronize()
        if world > 1 or one_rank:
            dist.barrier()
        return time.perf_counter() - t0

    # ---- timed region: exactly K steps ----
    elapsed = timed(args.steps)
    if world > 1 or one_rank:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu",
                         dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    records = gatherer.records_gathered if gatherer is not None else 0
    # host side of the step on THIS rank (the only new bottleneck a full node adds: 8 ranks'
    # launch threads): time spent submitting a step vs the step itself, slowest rank
    host_ms = 1e3 * host_submit[0] / max(1, host_submit[1])
    if world > 1:
        t = torch.tensor([host_ms], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        host_ms = float(t.item())

    # ---- the pipelined schedule only reorders launches: for EVERY image of the pool the
    # pipelined result must equal, bit for bit, what one eager single-stream call gives ----
    pipeline_check = None
    if rank == 0 and engine is not None and world == 1:
        n_img = len(pool) if args.path == "image" else 1
        got = []
        on_result[0] = lambda res: got.append(
            [[t.clone() for t in (r[1], r[7], r[4])] for r in res])
        nstep[0] = 0
        for _ in range(n_img):
            step()
        drain()
        on_result[0] = None
        torch.cuda.synchronize()
        g0, gb = head.use_graphs, getattr(backbone, "use_graphs", False)
        head.use_graphs = False
        if backbone is not None:
            backbone.use_graphs = False
        same = len(got) == n_img
        for i in range(n_img):
            ref = head.simple_test_bboxes(
                backbone(pool[i], slot=7) if backbone is not None else feats, metas)
            torch.cuda.synchronize()
            same &= all(torch.equal(a, b) for rg, rr in zip(got[i], ref)
                        for a, b in zip(rg, (rr[1], rr[7], rr[4])))
        head.use_graphs = g0
        if backbone is not None:
            backbone.use_graphs = gb
        pipeline_check = ("labels / rel_dists / pan_img of all %d distinct images are bitwise "
                          "the eager single-stream results" % n_img if same else "MISMATCH")
        if not same:
            raise SystemExit("a pipelined result differs from the eager single-stream result")

    # ---- secondary, headline-adjacent: the WHOLE of PSGTr.simple_test (psgtr.py:148-156):
    # the same steps plus the panoptic-loop status check and triplet2Result's device -> host
    # copy of every field (:15-51; the 2R x H0 x W0 bool masks, 49 MB per image, travel as
    # 6 MB of bits) into a ring of pinned host buffers, Results built on the host for every
    # image ----
    simple_test = None
    if rank == 0 and world == 1 and engine is not None and args.path == "image":
        from pairnet_amd import ResultStreamer
        streamer = ResultStreamer(head, ring=args.depth + 2)
        made = [0, None]

        def to_host(res):
            if len(streamer) >= streamer.ring - 1:
                made[1] = streamer.pop()
                made[0] += len(made[1])
            streamer.push(res, engine)
        on_result[0] = to_host

        def finish():
            while len(streamer):
                made[1] = streamer.pop()
                made[0] += len(made[1])
        for _ in range(2 * args.depth):
            step()
        drain()
        finish()
        made[0] = 0
        n = min(args.steps, 100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        drain()
        finish()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        on_result[0] = None
        r0 = made[1][0]
        simple_test = {
            "images_per_s": B * n / dt, "ms_per_step": 1e3 * dt / n, "steps": n,
            "results_built": made[0],
            "result_bytes_per_image": int(sum(getattr(r0, k).nbytes for k in (
                "refine_bboxes", "labels", "rel_pair_idxes", "rel_dists", "rel_labels",
                "pan_results", "masks"))),
            "d2h_bytes_per_image": int(sum(
                h.numel() * h.element_size() for e in streamer.entries[:1] if e is not None
                for host in e["host"] for h in host if isinstance(h, torch.Tensor))) // B,
            "what": "PSGTr.simple_test per step: image -> backbone -> head -> get_bboxes -> "
                    "panoptic status check -> triplet2Result (every field to pinned host memory "
                    "behind the image's query chain, the 2R x H0 x W0 bool masks as bits packed "
                    "on the device and expanded by host threads; Result objects built per image, "
                    "same fields / dtypes as the reference's; arrays are views of a %d-entry "
                    "ring)" % streamer.ring}
        del streamer

    # ---- secondary: the PRODUCT's distributed test loop on this rank's GPU --
    # `dist.multi_gpu_test` (the reference's multi_gpu_test + collect_results,
    # tools/test.py:256-267): shard -> PSGTr.stream_triplets -> records packed on the producing
    # chain stream -> delayed ring all-gather -> dataset-order records, on the detector built
    # from this run's backbone + head and its calibrated pipeline.  What the loop costs on top
    # of the bare step() loop above is the difference to the headline. ----
    product_loop = None
    if rank == 0 and world == 1 and engine is not None and args.path == "image" and B == 1 \
            and args.head == "pairnet" and not args.no_extras:
        from pairnet_amd import PSGTr
        from pairnet_amd.dist import multi_gpu_test
        det = PSGTr.from_parts(backbone, head)
        det._pipes = {args.depth: engine}       # (this run's calibrated stream placement)
        det._calibrated = {args.depth}
        n = min(args.steps, 100)
        data = [(pool[i % len(pool)], metas) for i in range(n)]
        multi_gpu_test(det, data[:2 * args.depth], depth=args.depth, force_collective=one_rank)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = multi_gpu_test(det, data, depth=args.depth, force_collective=one_rank)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        product_loop = {"images_per_s": n / dt, "ms_per_image": 1e3 * dt / n, "images": n,
                        "records": list(got["records"].shape), "collectives": got["collectives"],
                        "what": "pairnet_amd.dist.multi_gpu_test(detector, dataset) on one rank: "
                                "every image through PSGTr.stream_triplets, its triplet record "
                                "packed on the chain stream and gathered `depth` steps later"}
        # BASELINE configs[2]'s per-GPU batch through the same product loop: two images per
        # step (`samples_per_gpu=2`, tools/test.py:202-214; `dist.collate` stacks them)
        multi_gpu_test(det, data[:4 * args.depth], depth=args.depth, force_collective=one_rank,
                       samples_per_gpu=2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got2 = multi_gpu_test(det, data, depth=args.depth, force_collective=one_rank,
                              samples_per_gpu=2)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        product_loop["samples_per_gpu_2"] = {
            "images_per_s": n / dt2, "ms_per_step": 2e3 * dt2 / n, "images": n,
            "collectives": got2["collectives"], "records": list(got2["records"].shape)}
        head.use_graphs = not args.no_graphs
        backbone.use_graphs = not args.no_graphs
        head.grid_reserve = backbone.grid_reserve = engine.grid_reserve

    # ---- secondary: the reference's REAL call pattern (tools/test.py:199-267 with
    # Resize(keep_ratio) + Pad(size_divisor=1), configs/mask2former/pairnet.py:310-321): one
    # image per step, a new tensor shape every few images.  >= 200 images whose ORIGINAL sizes
    # are drawn from the COCO size histogram (13 distinct padded shapes, 16 original sizes,
    # seeded random order) through `dist.multi_gpu_test`: cold pass (every shape's first
    # sight inside the timed region), then a second pass; every record is checked bit for
    # bit against an eager single-stream call on that image alone. ----
    shape_mix = None
    if product_loop is not None and (H, W) == (800, 1333):
        from pairnet_amd.dist import multi_gpu_test, unpack_triplets
        from pairnet_amd.preprocess import rescale_size
        from pairnet_amd.head import CrossHead2 as _H
        ORIG = [((480, 640), 22), ((427, 640), 15), ((640, 480), 6), ((426, 640), 6),
                ((375, 500), 5), ((428, 640), 3), ((640, 427), 3), ((425, 640), 2),
                ((333, 500), 2), ((360, 640), 2), ((500, 375), 2), ((424, 640), 2),
                ((612, 612), 2), ((640, 640), 1), ((512, 640), 1), ((640, 426), 1)]
        n_mix = max(len(ORIG), args.mix_images)
        gm = torch.Generator().manual_seed(77)
        wts = torch.tensor([float(wt) for _, wt in ORIG])
        draw = torch.multinomial(wts, n_mix, replacement=True, generator=gm).tolist()
        draw[:len(ORIG)] = list(range(len(ORIG)))       # (every size at least once)
        perm = torch.randperm(n_mix, generator=gm).tolist()
        draw = [draw[i] for i in perm]
        imgs_by_shape, items = {}, []
        for k, oi in enumerate(draw):
            (h0, w0), _ = ORIG[oi]
            hn, wn = rescale_size(h0, w0, (1333, 800))
            pair = imgs_by_shape.setdefault((hn, wn), [
                torch.randn(1, 3, hn, wn, generator=gm).to(dev) for _ in range(2)])
            meta = dict(img_shape=(hn, wn, 3), ori_shape=(h0, w0, 3), pad_shape=(hn, wn, 3),
                        scale_factor=[wn / w0, hn / h0, wn / w0, hn / h0])
            items.append((pair[k % 2], [meta], (hn, wn), k % 2, (h0, w0)))
        data = [(im, m) for im, m, _, _, _ in items]
        det.reserve([(800, 1333), (1333, 800)], depth=args.depth,
                    orig_sizes=[sz for sz, _ in ORIG])
        arenas = list(head._arenas.values()) + list(head._post_arenas.values()) + \
            list(backbone._arenas.values())
        grows0 = sum(a.grows for a in arenas)
        caps0, ev0 = _H.captures, head._plans.evictions + backbone._plans.evictions
        recap0 = getattr(head, "recaptures", 0)
        torch.cuda.synchronize()
        mem0 = torch.cuda.memory_allocated()
        t0 = time.perf_counter()
        cold = multi_gpu_test(det, data, depth=args.depth, force_collective=one_rank)
        torch.cuda.synchronize()
        dt_cold = time.perf_counter() - t0
        mem_cold = torch.cuda.memory_allocated()
        caps1 = _H.captures
        t0 = time.perf_counter()
        warm = multi_gpu_test(det, data, depth=args.depth, force_collective=one_rank)
        torch.cuda.synchronize()
        dt_warm = time.perf_counter() - t0
        mem1 = torch.cuda.memory_allocated()
        # per-shape steady state: each padded shape alone, 24 images after its own warm-up
        alone_ms = {}
        for shp, pair in imgs_by_shape.items():
            (h0, w0) = next(o for _, _, s_, _, o in items if s_ == shp)
            m = next(m for _, m, s_, _, _ in items if s_ == shp)
            one = [(pair[i % 2], m) for i in range(24)]
            multi_gpu_test(det, one[:8], depth=args.depth, force_collective=one_rank)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            multi_gpu_test(det, one, depth=args.depth, force_collective=one_rank)
            torch.cuda.synchronize()
            alone_ms[shp] = 1e3 * (time.perf_counter() - t0) / len(one)
        ideal = sum(alone_ms[s_] for _, _, s_, _, _ in items) / 1e3
        # every record of both passes == an eager single-stream call on that image alone
        head.use_graphs = backbone.use_graphs = False
        head.grid_reserve = backbone.grid_reserve = 0
        want, same = {}, torch.equal(cold["records"], warm["records"])
        for k, (im, m, shp, which, orig) in enumerate(items):
            if (shp, which, orig) not in want:
                r = head.simple_test_bboxes(backbone(im, slot=7), m)[0]
                sub, obj = head.pair_positions()
                from pairnet_amd.dist import pack_triplets
                want[(shp, which, orig)] = pack_triplets(r[1], r[7], sub[0], obj[0]).clone()
            same &= torch.equal(cold["records"][k].to(dev), want[(shp, which, orig)])
        head.use_graphs = not args.no_graphs
        backbone.use_graphs = not args.no_graphs
        head.grid_reserve = backbone.grid_reserve = engine.grid_reserve
        if not same:
            raise SystemExit("shape_mix_product_loop: a record differs from the eager "
                             "single-shape result")
        shape_mix = {
            "images": n_mix, "padded_shapes": sorted("%dx%d" % s_ for s_ in imgs_by_shape),
            "original_sizes": len(ORIG),
            "images_per_s": n_mix / dt_cold, "ms_per_image": 1e3 * dt_cold / n_mix,
            "second_pass_images_per_s": n_mix / dt_warm,
            "per_shape_alone_ms": {"%dx%d" % k_: round(v, 3) for k_, v in sorted(alone_ms.items())},
            "ideal_images_per_s": n_mix / ideal,
            "frac_of_ideal": ideal / dt_cold, "second_pass_frac_of_ideal": ideal / dt_warm,
            "arena_grows_in_run": sum(a.grows for a in arenas) - grows0,
            "plan_evictions_in_run": head._plans.evictions + backbone._plans.evictions - ev0,
            "graph_captures_first_pass": caps1 - caps0,
            "graph_captures_second_pass": _H.captures - caps1,
            "stage_graph_recaptures_in_run": getattr(head, "recaptures", 0) - recap0,
            "arena_bytes": head.arena_bytes() + backbone.arena_bytes(),
            # (the first pass may also RELEASE memory: plans of earlier legs leave the LRU;
            # what must not happen is growth with the number of images / shapes seen)
            "device_bytes_growth_first_pass": int(mem_cold - mem0),
            "device_bytes_growth_second_pass": int(mem1 - mem_cold),
            "records_bitwise_eager_single_shape": bool(same),
            "what": "pairnet_amd.dist.multi_gpu_test over %d images of %d padded shapes "
                    "(original sizes from the COCO histogram, seeded order): first pass incl. "
                    "every shape's first sight, then a second pass; `ideal` = each image at "
                    "the steady-state rate of its own shape run alone; graphs are captured at quiet "
                    "points only (no other stream executing), so shapes first met in flight run "
                    "eagerly; a recapture would be a capture of a (shape, slot, stage) that had "
                    "a graph before" % (
                        n_mix, len(imgs_by_shape))}
        del imgs_by_shape, items, data

    # ---- secondary: the head alone on the resident pyramid (round 1's headline) ----
    head_only = None
    if args.path == "image" and world == 1:
        # (its own pipeline: without a backbone in front one stage-A stream and two chain
        # streams are the faster schedule for the head alone)
        n = min(args.steps, 100)
        he = engine
        if engine is not None:
            he = PipelinedHead(head, depth=3, a_streams=1, grid_trim=32)
            for _ in range(4):
                he.submit(feats, metas)
            he.flush()
            he.calibrate(feats, metas)

        def head_steps(k):
            for _ in range(k):
                if he is None:
                    head.simple_test_bboxes(feats, metas)
                else:
                    he.submit(feats, metas)
            if he is not None:
                he.flush()
        head_steps(4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        head_steps(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        head_only = {"images_per_s": B * n / dt, "ms_per_step": 1e3 * dt / n, "steps": n,
                     "what": "%s.simple_test_bboxes on the feature pyramid resident in HBM (no "
                             "backbone; 3-stream pipeline: one stage-A stream, two chain "
                             "streams)" % type(head).__name__}
        if engine is not None:   # (the head-only pipeline set its own reserve on the head)
            head.grid_reserve = engine.grid_reserve

    # ---- secondary: from the DECODED image (uint8 HWC BGR resident in HBM): the reference's
    # test pipeline (resize keep-ratio / normalise / pad, pairnet.py:310-331) as one kernel in
    # front of the backbone, on the same stream ----
    from_decoded = None
    if args.path == "image" and world == 1 and B == 1:
        from pairnet_amd import TestPipeline
        from pairnet_amd.preprocess import rescale_size
        h0, w0 = round(H / sf), round(W / sf)
        if rescale_size(h0, w0, (1333, 800)) == (H, W):
            pipe = TestPipeline(device=dev)
            raw = torch.randint(0, 256, (h0, w0, 3), generator=g, dtype=torch.uint8).to(dev)

            imgs = [img] + [torch.empty_like(img) for _ in range(
                len(engine.streams_a) - 1 if engine is not None else 0)]

            def raw_step():
                sl = engine.count % len(engine.streams_a) if engine is not None else 0
                with torch.cuda.stream(engine.streams_a[sl] if engine is not None
                                       else torch.cuda.current_stream()):
                    _, m = pipe(raw, out=imgs[sl])
                    if engine is None:
                        return head.simple_test_bboxes(backbone(imgs[sl]), m)
                    return engine.submit(backbone(imgs[sl], slot=sl), m)
            n = min(args.steps, 100)
            for _ in range(4):
                raw_step()
            drain()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                raw_step()
            drain()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            from_decoded = {"images_per_s": n / dt, "ms_per_step": 1e3 * dt / n, "steps": n,
                            "what": "uint8 %dx%dx3 BGR image in HBM -> TestPipeline (resize to "
                                    "%dx%d, normalise, pad; one HIP kernel) -> %s -> head"
                                    % (h0, w0, H, W, bname)}
            if engine is not None and rank == 0:
                # host to host, PCIe both ways (never `value`): the decoded uint8 image in
                # PINNED HOST memory -> H2D (0.74 MB) -> TestPipeline -> backbone -> head ->
                # get_bboxes -> ResultStreamer -> Result objects on the host
                from pairnet_amd import ResultStreamer
                raw_host = raw.cpu().pin_memory()
                raws = [torch.empty_like(raw) for _ in imgs]
                streamer = ResultStreamer(head, ring=args.depth + 2)
                built = [0]

                def take(res):
                    if len(streamer) >= streamer.ring - 1:
                        built[0] += len(streamer.pop())
                    streamer.push(res, engine)

                def host_step():
                    sl = engine.count % len(engine.streams_a)
                    with torch.cuda.stream(engine.streams_a[sl]):
                        raws[sl].copy_(raw_host, non_blocking=True)
                        _, m = pipe(raws[sl], out=imgs[sl])
                        res = engine.submit(backbone(imgs[sl], slot=sl), m)
                        if res is not None:
                            take(res)

                def host_drain():
                    with torch.cuda.stream(engine.streams_a[0]):
                        for res in engine.flush():
                            take(res)
                    while len(streamer):
                        built[0] += len(streamer.pop())
                for _ in range(2 * args.depth):
                    host_step()
                host_drain()
                built[0] = 0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    host_step()
                host_drain()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                from_decoded["host_to_host"] = {
                    "images_per_s": n / dt, "ms_per_step": 1e3 * dt / n, "steps": n,
                    "results_built": built[0],
                    "what": "PCIe-inclusive: the decoded uint8 image starts in pinned host memory "
                            "(H2D 0.74 MB per image) and every Result field ends in host memory "
                            "(D2H 8.1 MB per image, masks as bits): TestPipeline -> backbone -> "
                            "head -> get_bboxes -> triplet2Result, pipelined"}
                streamer.close()
                del streamer
            img.copy_(img_cpu)            # (the legs below run on the seeded tensor again)

    # ---- roofline leg: the same step again, eagerly on one stream, with HIP events
    # (recorded on the launching stream) around every GEMM / conv / MSDA launch.  Kept out
    # of the timed region above because ~150 event pairs per step cost host time there. ----
    timer = dominant = prof = None
    nprof = min(args.steps, 10)
    if rank == 0:
        head.use_graphs = False   # events need eager, single-stream launches
        if backbone is not None:
            backbone.use_graphs = False
        hip.TIMER = timer = hip.KernelTimer()
        for _ in range(nprof):
            head.simple_test_bboxes(backbone(img) if backbone is not None else feats, metas)
        prof = timer.summary()
        # the dominant kernel's launches by shape (= by flop count and byte count)
        by_shape = {}
        for name, flops, nbytes, s_ev, e_ev in timer.records:
            a = by_shape.setdefault((name, flops, nbytes), [0, 0.0])
            a[0] += 1
            a[1] += s_ev.elapsed_time(e_ev)
        hip.TIMER = None
        head.use_graphs = not args.no_graphs
        if backbone is not None and not swin:
            backbone.use_graphs = not args.no_graphs
        dominant = max(prof, key=lambda k: prof[k]["ms"]) if prof else None
    # (the collector stays off through the extras below: their timed loops are as short as
    # the headline's; it is re-enabled for the CPU baseline)

    out = None
    if rank == 0:
        images = world * B * args.steps
        if sibling:
            what = ("SIBLING HEAD (not the headline): %s on the same pixel decoder + 9-layer "
                    "masked decoder" % type(head).__name__)
        else:
            what = ("Pair-Net %s + Mask2Former (CrossHead2: pixel decoder -> 9-layer masked "
                    "decoder -> PPN / Matrix Learner / top-k -> 6-layer relation decoder -> "
                    "get_bboxes)" % (bname or "head"))
        what += (", image tensor -> backbone -> head -> triplets" if args.path == "image" else
                 ", HEAD ONLY on a feature pyramid resident in HBM")
        out = {
            "metric": "images/sec (whole node), 100-query 800x1333, 1/2/4/8 MI355X",
            "value": images / elapsed, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": what + ", %d object / %d relation queries, channels %s, bs=%d per "
                            "GPU, %dx%d, %s resident in HBM, random-init weights"
                            % (head.num_obj_query, head.num_rel_query, list(chans), B, H, W,
                               "normalised image tensor" if args.path == "image"
                               else "feature pyramid"),
                "path": args.path, "backbone": bname,
                "distinct_input_images": len(pool) if args.path == "image" else 1,
                "attention_mask_order": {
                    "reference": "the reference's: bilinear resize of full-size mask logits, "
                                 "computed at the 4 logits per key the resize reads",
                    "dense": "the reference's: all full-size mask logits, then the resize",
                    "resampled": "shortcut: logits against the mask feature resampled once per "
                                 "level"}[args.mask_order],
                "gemm_arithmetic": GEMM_ARITHMETIC[args.gemm_arithmetic],
                "global_batch": world * B, "per_gpu_batch": B, "image": [H, W],
                "stream_placement_calibration_ms": calibration,
                "parallelism": "dp%d" % world,
                "schedule": ("eager" if args.no_graphs else "hipGraph replay per stage") + (
                    ", single stream" if args.no_pipeline else
                    ", %d-stream pipeline: backbone + stage A of consecutive batches alternate "
                    "between %d stream(s), their query chains run on %d more"
                    % (args.depth, args.a_streams, args.depth - args.a_streams)),
                "collective": ("RCCL all-gather of triplet records, once per step"
                               if (world > 1 or one_rank) and backend == "nccl" else
                               "gloo all-gather (functional check)" if world > 1 else "none")},
            "rccl_ranks": world if backend == "nccl" else 0,
            "dist_backend": backend if world > 1 or one_rank else None,
            "triplet_records_gathered": records,
            "triplet_record_bytes": 4 * gatherer.L if gatherer is not None else None,
            "pipeline_check": pipeline_check,
            "host": {"submit_ms_per_step_max_over_ranks": host_ms,
                     "submit_fraction_of_step": host_ms / (1e3 * elapsed / args.steps),
                     "numa_binding": numa_bound,
                     "what": "host time inside step() per step (graph replays + eager launches "
                             "+ the per-step collective's enqueue) on the slowest rank; ranks "
                             "are bound to the NUMA node of their GPU when --gpus > 1 "
                             "(dist.bind_to_gpu_numa; PAIRNET_NUMA_BIND=1 forces it on one GPU)"},
        }
        if simple_test is not None:
            out["simple_test_incl_result_d2h"] = simple_test
        if product_loop is not None:
            out["multi_gpu_test_product_loop"] = product_loop
        if shape_mix is not None:
            out["shape_mix_product_loop"] = shape_mix
        if head_only is not None:
            out["head_only"] = head_only
        if from_decoded is not None:
            out["from_decoded_image"] = from_decoded
        if timer and dominant:
            def roof(name):
                agg = prof[name]
                sec = agg["ms"] * 1e-3
                if name in HBM_KERNELS:
                    ach, peak, unit, bound = agg["bytes"] / sec / 1e9, PEAK_HBM_GBS, "GB/s", "hbm"
                elif name in BF16X3_KERNELS:
                    # issued work: six bf16 MFMA products per fp32 multiply-add, against the
                    # dense bf16 MFMA peak
                    ach, peak, unit = 6.0 * agg["flops"] / sec / 1e12, PEAK_BF16_MFMA_TFLOPS, "TFLOP/s"
                    bound = "mfma"
                else:
                    ach, peak, unit = agg["flops"] / sec / 1e12, PEAK_F32_MFMA_TFLOPS, "TFLOP/s"
                    bound = "mfma"
                traffic, traffic_src, mfma_util = None, None, None
                try:  # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE
                    # doubled per MI355X_MICROARCH.md + WRITE_SIZE), same workload, same kernel
                    src = latest_pmc()
                    pmc = json.load(open(src))
                    hits = [v for v in pmc["kernels"].values()
                            if v.get("bench_name") == name or
                            (name in HBM_KERNELS and str(v.get("bench_name")).startswith(name))]
                    if hits:   # template variants of one kernel: launch-weighted mean
                        n = sum(v["launches_profiled"] for v in hits)
                        traffic = int(sum(v["hbm_bytes_per_launch"] * v["launches_profiled"]
                                          for v in hits) / n)
                        traffic_src = os.path.relpath(src, ROOT)
                        sq = [v for v in hits if "sq" in v]
                        if sq:   # matrix-pipe busy cycles / (4 x CU-busy cycles), PMC pass
                            mfma_util = sum(v["sq"]["mfma_util"] * v["launches_profiled"]
                                            for v in sq) / sum(v["launches_profiled"] for v in sq)
                except (OSError, ValueError, KeyError, TypeError):
                    pass
                extra = {}
                if name in BF16X3_KERNELS:
                    extra = {"mfma_dtype": "bf16 (six products per fp32 multiply-add, fp32 accumulate)",
                             "fp32_equivalent_tflops": agg["flops"] / sec / 1e12,
                             "fp32_equivalent_over_fp32_mfma_peak": agg["flops"] / sec / 1e12 / PEAK_F32_MFMA_TFLOPS}
                return {
                    **extra,
                    "kernel": name, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                    "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src,
                    "mfma_util_pmc": mfma_util,
                    "algorithmic_bytes_per_launch": agg["bytes"] / agg["launches"],
                    "launches_per_step": agg["launches"] // nprof,
                    "avg_launch_us": 1e3 * agg["ms"] / agg["launches"],
                    "ms_per_step": agg["ms"] / nprof,
                    "measured": "HIP events around each launch on the launching stream, %d "
                                "eager single-stream steps right after the timed region" % nprof}

            out["roofline"] = roof(dominant)
            if out["roofline"]["bound"] == "mfma":
                # the same kernel per problem shape, largest share of its time first: the
                # fraction above is a launch mix (big encoder GEMMs, mid-size backbone 1x1
                # convolutions whose tile counts quantise badly on 1024 resident workgroups)
                shapes = sorted(((k, v) for k, v in by_shape.items() if k[0] == dominant),
                                key=lambda kv: -kv[1][1])[:6]
                out["roofline"]["by_shape"] = [
                    {"gflop_per_launch": k[1] * 1e-9, "launches_per_step": v[0] // nprof,
                     "avg_launch_us": 1e3 * v[1] / v[0],
                     "tflops": k[1] / (v[1] / v[0] * 1e-3) / 1e12,
                     "frac": k[1] / (v[1] / v[0] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                     "share_of_kernel_time": v[1] / prof[dominant]["ms"]} for k, v in shapes]
            # round 4: twelve of the dominant kernel's launches per image (the encoder's two
            # N = 256 Linears, among them FFN-2, its best-running shape) moved into the
            # row-owning Linear + residual + LayerNorm kernel: reported beside it, and both
            # together as the fp32-MFMA GEMM work of the step
            if "k_gemm_rowln" in prof:
                out["roofline_linear_res_ln"] = roof("k_gemm_rowln")
            # round 6: the encoder's GEMMs on the bf16 matrix pipe (both instantiations)
            for k in BF16X3_KERNELS:
                if k in prof and k != dominant:
                    out["roofline_" + k.replace("<", "_").replace(">", "")] = roof(k)
            f32g = [k for k in prof if k.startswith(("k_gemm_tile", "k_gemm_rowln", "k_gemm_group",
                                                     "k_gemm_stencil"))]
            gemms = f32g + [k for k in prof if k in BF16X3_KERNELS]
            if gemms:
                fl = sum(prof[k]["flops"] for k in gemms)
                ms = sum(prof[k]["ms"] for k in gemms)
                # all GEMM kernels of the step in fp32-EQUIVALENT work (2 M N K per GEMM whatever
                # instruction carries it) over their summed time; the fp32 MFMA peak beside it is
                # the roof of the fp32 kernels only -- kept as the yardstick of rounds 1-5
                out["roofline_all_gemm_kernels"] = {
                    "kernels": sorted(gemms), "bound": "mfma", "unit": "TFLOP/s fp32-equivalent",
                    "achieved": fl / (ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                    "frac": fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                    "ms_per_step": ms / nprof,
                    "launches_per_step": sum(prof[k]["launches"] for k in gemms) // nprof,
                    "fp32_mfma_kernels": {
                        "achieved": sum(prof[k]["flops"] for k in f32g) /
                        (sum(prof[k]["ms"] for k in f32g) * 1e-3) / 1e12 if f32g else None,
                        "ms_per_step": sum(prof[k]["ms"] for k in f32g) / nprof}}
            # the north star's other named kernel: achieved HBM GB/s of the deformable sampling
            msda = [k for k in prof if k.startswith("k_msda")]
            if msda:
                r = out["roofline_deformable_sampling"] = roof(msda[0])
                # HBM is the wrong roof for a gather whose value rows are L2-resident (each
                # XCD samples its own band): the roof it is under is the rate the vector L1 /
                # L2 deliver for this access pattern, measured here by a bare-gather probe in
                # the same run (never a constant).  One layer passes N x 8 heads x 12 (level,
                # point) x 4 taps x 128 B through the L1.
                try:
                    probe = gather_roof(dev, hip)
                    l1_bytes = 21950.0 * (H * W) / (800 * 1333) * 8 * 12 * 4 * 128 \
                        if (H, W) != (800, 1333) else 21950.0 * 8 * 12 * 4 * 128
                    ach = l1_bytes / (r["avg_launch_us"] * 1e-6) / 1e12
                    r.update({
                        "gather_bytes_per_launch": l1_bytes, "gather_achieved": ach,
                        "gather_unit": "TB/s", "gather_peak": probe[2048],
                        "frac_of_gather_peak": ach / probe[2048],
                        "gather_probe_tbs_by_window_kb": {str(k): v for k, v in probe.items()},
                        "gather_peak_source": "k_gather_probe in this run: 8 lanes x 16 B per "
                                              "random 128-byte line, 12 lines in flight per lane "
                                              "group, 2 MB window (one XCD band of the value map)"})
                except Exception as e:      # (a probe failure must not lose the bench line)
                    r["gather_peak"] = None
                    r["gather_probe_error"] = repr(e)
            # ... and its attention GEMMs (QK^T / PV on the fp32 MFMA, flash-style): small
            # latency-bound launches of the query chain; matrix-pipe utilisation from the PMC pass
            if "k_attn_chunk" in prof:
                out["roofline_attention"] = roof("k_attn_chunk")
            out["kernel_profile"] = {
                k: {"ms_per_step": v["ms"] / nprof, "launches_per_step": v["launches"] // nprof,
                    "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0.0,
                    "gbs": v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] else 0.0}
                for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}

    # (single-GPU runs only: the extras re-enter step(), whose triplet all-gather is a
    # collective the other ranks would not be in)
    if rank == 0 and world == 1 and not args.no_extras:
        def timeit(fn, n=5):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t) / n

        def whole():
            return head.simple_test_bboxes(
                backbone(img) if backbone is not None else feats, metas)
        head.use_graphs = False
        if backbone is not None:
            backbone.use_graphs = False
        out["latency_ms_single_stream_eager"] = timeit(whole, 10)
        outs = head.forward(feats, metas)
        out["breakdown_ms"] = {
            "forward": timeit(lambda: head.forward(feats, metas)),
            "get_bboxes": timeit(lambda: head.get_bboxes(*outs, metas))}
        if backbone is not None:
            out["breakdown_ms"]["backbone_%s_native_fp32_mfma" % bname] = min(
                timeit(lambda: backbone(img), 10), timeit(lambda: backbone(img), 10))
        head.use_graphs = not args.no_graphs
        if backbone is not None and not swin:
            backbone.use_graphs = not args.no_graphs
        out["latency_ms_single_stream_graphs"] = timeit(whole, 10)
        if args.head == "pairnet" and hasattr(head, "loss"):
            # SURVEY 8 f4 (forward slice): the loss VALUES of CrossHead2.loss on this step's
            # outputs with a synthetic ground truth (12 instance masks at half the batch
            # tensor's size, 10 relations per image, the config's 12544 sampled points)
            try:
                g = torch.Generator().manual_seed(5)
                G, T = 12, 10
                Hh, Wh = (int(img.shape[2]) if img is not None else 4 * feats[0].shape[2]) // 2, \
                    (int(img.shape[3]) if img is not None else 4 * feats[0].shape[3]) // 2
                gt_masks = [(torch.rand(G, Hh // 8, Wh // 8, generator=g) > 0.7).to(dev)
                            .repeat_interleave(8, 1).repeat_interleave(8, 2).contiguous()
                            for _ in range(B)]
                gt_labels = [torch.randint(0, head.num_classes, (G,), generator=g) for _ in range(B)]
                gt_rels = [torch.stack([torch.randint(0, G, (T,), generator=g),
                                        torch.randint(0, G, (T,), generator=g),
                                        torch.randint(1, head.num_relations + 1, (T,), generator=g)], 1)
                           for _ in range(B)]
                lossf = lambda: head.loss(*outs, gt_rels, None, gt_labels, gt_masks, metas)
                vals = lossf()
                out["loss_forward"] = {
                    "ms_per_batch": timeit(lossf, 10), "images": B,
                    "values": {k: float(v) for k, v in vals.items()},
                    "what": "CrossHead2.loss on the forward outputs (values only): point "
                            "sampling of 100 mask logit maps + %d ground-truth masks at 12544 "
                            "points, the two match-cost matrices on the GPU, both Hungarian "
                            "assignments with scipy on the host (as the reference), Seesaw / CE / "
                            "BCE(pos_weight) reductions on the GPU" % G}
            except Exception as e:      # noqa: BLE001 -- an extra, never the headline
                out["loss_forward"] = repr(e)
        # SURVEY 8 f1 (ground-truth side): panoptic PNG -> per-segment masks, one kernel
        try:
            from pairnet_amd import hip as _hip
            g = torch.Generator().manual_seed(7)
            Hg, Wg, Gs = 480, 640, 20
            ids = torch.randperm(2 ** 24 - 2, generator=g)[:Gs].to(torch.int32) + 1
            grid = torch.randint(0, Gs, (Hg // 8, Wg // 8), generator=g) \
                .repeat_interleave(8, 0).repeat_interleave(8, 1)
            seg = ids.long()[grid]
            rgb = torch.stack([seg % 256, (seg // 256) % 256, seg // 65536], -1).to(torch.uint8).to(dev)
            ids_d, cats_d = ids.to(dev), torch.arange(Gs, dtype=torch.int32, device=dev)
            gm = torch.empty((Gs, Hg, Wg), dtype=torch.uint8, device=dev)
            sem = torch.empty((Hg, Wg), dtype=torch.int32, device=dev)
            run = lambda: _hip.pan_masks(rgb, ids_d, cats_d, gm, sem)
            for _ in range(5):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(200):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / 200
            nbytes = Hg * Wg * (3 + Gs + 4)
            ok = bool(torch.equal(gm.cpu(), (seg[None] == ids.long()[:, None, None]).to(torch.uint8)))
            # the same kernel where it is bandwidth- rather than launch-sized
            Hb, Wb, Gb = 2048, 2048, 64
            rgb_b = torch.randint(0, 256, (Hb, Wb, 3), generator=g, dtype=torch.uint8).to(dev)
            ids_b = torch.arange(1, Gb + 1, dtype=torch.int32, device=dev)
            gm_b = torch.empty((Gb, Hb, Wb), dtype=torch.uint8, device=dev)
            big = lambda: _hip.pan_masks(rgb_b, ids_b, None, gm_b)
            for _ in range(3):
                big()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                big()
            e1.record()
            torch.cuda.synchronize()
            us_b = 1e3 * e0.elapsed_time(e1) / 20
            nb_b = Hb * Wb * (3 + Gb)
            out["ground_truth_masks"] = {
                "large_2048x2048x64": {"us": us_b, "gbs": nb_b / us_b * 1e-3,
                                       "frac_of_hbm": nb_b / us_b * 1e-3 / 8000.0},
                "kernel": "k_pan_masks", "us_per_image": us, "bytes_per_image": nbytes,
                "gbs": nbytes / us * 1e-3, "frac_of_hbm": nbytes / us * 1e-3 / 8000.0,
                "equals_rgb2id_compare": ok,
                "what": "480x640 panoptic PNG (RGB, resident in HBM) -> 20 segment masks + semantic "
                        "map (psg.py:354-372, loading.py:128-147), back-to-back launches incl. "
                        "launch overhead: a 8 MB problem is launch-latency sized"}
        except Exception as e:      # noqa: BLE001 -- an extra, never the headline
            out["ground_truth_masks"] = repr(e)
        if engine is not None and args.gemm_arithmetic == "bf16x3":
            # the same pipelined steps with the encoder on the exact-fp32 MFMA kernels of rounds
            # 1-5 (one flag away: head.gemm_arithmetic = "fp32")
            head.gemm_arithmetic = "fp32"
            if hasattr(backbone, "gemm_arithmetic"):
                backbone.gemm_arithmetic = "fp32"
            for _ in range(2 * args.depth):      # (graphs are re-captured for the new setting)
                step()
            drain()
            n = min(args.steps, 40)
            dt = timed(n)
            head.gemm_arithmetic = "bf16x3"
            if hasattr(backbone, "gemm_arithmetic"):
                backbone.gemm_arithmetic = "bf16x3"
            for _ in range(2 * args.depth):
                step()
            drain()
            out["fp32_mfma_encoder"] = {
                "images_per_s": B * n / dt, "ms_per_step": 1e3 * dt / n, "steps": n,
                "what": GEMM_ARITHMETIC["fp32"] + " (head.gemm_arithmetic = 'fp32'); the headline "
                        "runs the encoder's GEMMs as " + GEMM_ARITHMETIC["bf16x3"]}
        if engine is not None and args.mask_order == "reference":
            # the same pipelined steps with the opt-in shortcut for the attention masks: the
            # mask feature is resampled to each level once per image and every layer's logits
            # are a Q x N_l GEMM against it (resize(me . MF) == me . resize(MF) up to fp32
            # re-association) instead of the reference's operation order of the headline
            head.exact_mask_order = False
            for _ in range(2 * args.depth):      # (graphs are re-captured for the new setting)
                step()
            drain()
            dt = timed(min(args.steps, 40))
            head.exact_mask_order = True
            for _ in range(2 * args.depth):
                step()
            drain()
            n = min(args.steps, 40)
            out["resampled_mask_feature_shortcut"] = {
                "images_per_s": B * n / dt, "ms_per_step": 1e3 * dt / n, "steps": n,
                "what": "headline schedule with exact_mask_order=False (opt-in): attention-mask "
                        "logits against the once-resampled mask feature instead of the "
                        "reference's order of operations"}
        if engine is not None and B == 1 and args.path == "image":
            # BASELINE configs[2] is 2 images per GPU (bs = 16 over 8 GPUs): the same schedule
            # with two distinct images per launch sequence (plans and graphs of their own)
            pool2 = [torch.cat([pool[i], pool[(i + 1) % len(pool)]], 0)
                     for i in range(0, len(pool), 2)]
            metas2, k2 = metas * 2, [0]

            def step2():
                sl = engine.count % len(engine.streams_a)
                with torch.cuda.stream(engine.streams_a[sl]):
                    engine.submit(backbone(pool2[k2[0] % len(pool2)], slot=sl), metas2)
                k2[0] += 1

            def run2(n):
                for _ in range(n):
                    step2()
                with torch.cuda.stream(engine.streams_a[0]):
                    engine.flush()
            run2(2 * args.depth)
            run2(4)
            torch.cuda.synchronize()
            n2 = min(args.steps, 40)
            t0 = time.perf_counter()
            run2(n2)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out["two_images_per_gpu"] = {
                "images_per_s": 2 * n2 / dt, "ms_per_step": 1e3 * dt / n2, "steps": n2,
                "what": "BASELINE configs[2]'s per-GPU batch: two distinct images per step "
                        "through the same pipeline (one launch per layer for both)"}
            del pool2
        # deformable sampling with learned-like offsets: default-init weights give every token
        # mmcv's +-1..4 px grid (a tight neighbourhood); the same kernel on the grid plus
        # N(0, 8 px) noise per offset shows how far the rate depends on that locality
        try:
            pl = head._plan(B, [tuple(f.shape[-2:]) for f in feats[:0:-1]],
                            tuple(feats[0].shape[-2:]), 0, getattr(head, "_feats_nhwc", False))
            voa = pl.VOA.clone()
            gen = torch.Generator(device=dev).manual_seed(7)
            voa[..., 256:448] += 8.0 * torch.randn(voa[..., 256:448].shape, device=dev, generator=gen)
            s_out = torch.empty_like(pl.S)
            run = lambda v: hip.msda(v, 544, v.view(-1)[256:], 544, s_out, pl.B, pl.shapes)
            t_init, t_spread = timeit(lambda: run(pl.VOA), 50), timeit(lambda: run(voa), 50)
            alg = 4.0 * pl.B * pl.SN * (256 + 8 * 3 * 4 * 3 + 256)
            out["deformable_sampling_offsets"] = {
                "init_grid": {"us": 1e3 * t_init, "GB/s": alg / t_init / 1e6},
                "init_grid_plus_N(0,8px)": {"us": 1e3 * t_spread, "GB/s": alg / t_spread / 1e6},
                "what": "pn_msda_f32 alone, back to back, on the last step's [value|offsets|"
                        "logits] buffer and on a copy with noise added to every offset"}
            del voa, s_out
        except Exception as e:  # pragma: no cover
            out["deformable_sampling_offsets"] = repr(e)
        # random weights make the attention masks ~50 % dense, a trained
        # head's are sparser -- the masked cross-attention of each level on the masks of the
        # last step and on a synthetic mask with 10 % foreground (every query attends one
        # rectangle of a tenth of the map, the rest is masked)
        if args.head == "pairnet":
            try:
                Q = head.num_obj_query
                res = {}
                for l, (h, wd) in enumerate(pl.shapes):
                    n = h * wd
                    nw = (n + 31) // 32
                    gen = torch.Generator(device=dev).manual_seed(11 + l)
                    lg = torch.randn(B * Q, n, device=dev, generator=gen)      # ~50 % masked
                    fg = torch.full((B * Q, h, wd), -1.0, device=dev)
                    rh, rw = max(1, int(round(h * 0.316))), max(1, int(round(wd * 0.316)))
                    ys = torch.randint(0, h - rh + 1, (B * Q,), generator=gen, device=dev).tolist()
                    xs = torch.randint(0, wd - rw + 1, (B * Q,), generator=gen, device=dev).tolist()
                    for r, (y0, x0) in enumerate(zip(ys, xs)):
                        fg[r, y0:y0 + rh, x0:x0 + rw] = 1.0
                    fg = fg.view(B * Q, n)
                    bits = torch.empty(B * Q * nw, dtype=torch.int32, device=dev)
                    rowall = torch.empty(B * Q, dtype=torch.int32, device=dev)
                    row = {"keys": n}
                    for name, logit in (("random_half", lg), ("foreground_10pct", fg)):
                        hip.mask_pack(logit, bits, rowall, B * Q, n)
                        t = timeit(lambda: hip.attention(
                            pl.Qp, 256, pl.Kp[l], 256, pl.Vp[l], 256, bits, rowall, pl.att, 256,
                            pl.scr, B, Q, n, 1.0 / math.sqrt(32.0)), 50)
                        row[name] = {"attendable_fraction": float((logit >= 0).float().mean()),
                                     "us": 1e3 * t}
                    res["level_%d" % l] = row
                    del lg, fg, bits, rowall
                res["what"] = ("pn_attention_f32 alone (Q = %d queries x 8 heads against one "
                               "level's keys), back to back, on a ~50 %% random mask and on a "
                               "mask with one 10 %% rectangle of foreground per query: key tiles "
                               "no query of a wave may attend to are skipped" % Q)
                out["attention_mask_density"] = res
            except Exception as e:  # pragma: no cover
                out["attention_mask_density"] = repr(e)
        # RCCL on this box: one rank is what a single GPU can run of the multi-GPU path --
        # communicator set-up and the bench's collectives (all-gather of the triplet records on
        # a side stream, MAX reduction, barrier) through the real library.  In a child process
        # with a time limit: nothing of the run above depends on it.
        if backend == "nccl":
            try:
                probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools",
                                     "rccl_single_rank_probe.py")
                cp = subprocess.run([sys.executable, probe], capture_output=True, text=True,
                                    timeout=120)
                line = [l for l in cp.stdout.splitlines() if l.startswith("RCCL_JSON ")]
                out["rccl_single_rank"] = (json.loads(line[-1][len("RCCL_JSON "):]) if line
                                           else "no result: " + cp.stderr[-300:])
            except Exception as e:  # pragma: no cover
                out["rccl_single_rank"] = repr(e)
        if backbone is not None and not swin:
            try:  # comparison leg: the same backbone through PyTorch-ROCm / MIOpen
                from tools.torch_resnet50 import ResNet50
                bb = ResNet50().to(dev)
                out["breakdown_ms"]["backbone_r50_torch_miopen"] = timeit(lambda: bb(img), 3)
                del bb
            except Exception as e:  # pragma: no cover
                out["breakdown_ms"]["backbone_torch_error"] = repr(e)

    if rank == 0 and world == 1 and not args.no_extras and not args.no_sub_legs and \
            args.head == "pairnet" and args.path == "image" and (H, W) == (800, 1333) and not swin:
        # BASELINE configs[3] ("Swin-L + Mask2Former, 200 object queries", one image per GPU) and
        # the box-trunk sibling, each as a short run of this script in a child process (its own
        # weights, plans and calibration; this process is idle meanwhile): driver-timed lines
        # for the configurations the headline does not cover
        def leg(extra, steps=16, warm=6):
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps),
                   "--warmup", str(warm), "--no-extras", "--no-cpu-baseline", "--no-sub-legs",
                   "--gemm-arithmetic", args.gemm_arithmetic] + extra
            t0 = time.perf_counter()
            try:
                cp = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
                line = [l for l in cp.stdout.splitlines() if l.startswith("{")]
                if not line:
                    return "no result (rc %d): %s" % (cp.returncode, cp.stderr[-400:])
                r = json.loads(line[-1])
                keep = ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "pipeline_check",
                        "roofline", "roofline_all_gemm_kernels")
                res = {k: r[k] for k in keep if k in r}
                res["workload"] = r["config"]["workload"]
                res["gemm_arithmetic"] = r["config"].get("gemm_arithmetic")
                res["child_process_s"] = time.perf_counter() - t0
                return res
            except Exception as e:      # noqa: BLE001 -- a secondary leg, never the headline
                return repr(e)
        torch.cuda.synchronize()
        out["swin_l_200q"] = leg(["--in-channels", "192,384,768,1536", "--queries", "200"])
        out["box_trunk"] = leg(["--head", "bbox"])
        # what one training iteration of the head behind the pixel decoder costs
        # (tools/train_step_probe.py in a child process: it re-homes the head's weights)
        try:
            t0 = time.perf_counter()
            cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_step_probe.py"), "10"],
                                capture_output=True, text=True, timeout=600)
            line = [l for l in cp.stdout.splitlines() if l.startswith("{")]
            out["training_step"] = json.loads(line[-1]) if line else \
                "no result (rc %d): %s" % (cp.returncode, cp.stderr[-400:])
            if isinstance(out["training_step"], dict):
                out["training_step"]["child_process_s"] = time.perf_counter() - t0
        except Exception as e:          # noqa: BLE001 -- a secondary leg, never the headline
            out["training_step"] = repr(e)

    gc.enable()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.head import OracleCrossHead2
        from oracle.baseline_head import OracleCrossHeadBaseline
        from oracle.psgtr_head2 import OraclePSGTrHead2
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") \
            else (os.cpu_count() or 1)
        oracle = {"pairnet": OracleCrossHead2, "baseline": OracleCrossHeadBaseline,
                  "psgtr2": OraclePSGTrHead2}[args.head](**cfg).eval()
        oracle.load_state_dict(head.state_dict())
        obb = None
        if backbone is not None and not swin:
            from oracle.backbone import OracleResNet50
            obb = OracleResNet50()
            obb.load_state_dict(backbone.state_dict())
        elif backbone is not None:
            from oracle.swin import OracleSwin
            scfg = swin_backbone_cfg(swin)
            obb = OracleSwin(**{k: scfg[k] for k in ("embed_dims", "depths", "num_heads",
                                                      "window_size") if k in scfg}).eval()
            obb.load_state_dict(backbone.state_dict())
        if feats_cpu is None and obb is None:
            feats_cpu = [f.cpu().contiguous() for f in feats]

        def cpu_pass():
            t = time.perf_counter()
            with torch.no_grad():
                f = [x.contiguous() for x in obb(img_cpu)] if obb is not None else feats_cpu
                oracle.simple_test_bboxes(f, metas)
            return time.perf_counter() - t
        # The thread count is chosen by timing the WHOLE pass at rising settings up to all
        # host cores.  torch's CPU kernels stop scaling long before 256 threads on these
        # shapes and then collapse (measured on the GPU box: 2.9 s per pass at 8 threads,
        # 5.0 s at 64, 141.6 s at all 256): the sweep stops once a pass is 1.5 x slower than
        # the best one, so that this leg stays within its 10-30 s budget.
        cands = sorted({n for n in (8, 16, 32, 64, 128, avail) if n <= avail})
        sweep = {}
        torch.set_num_threads(cands[0])
        cpu_pass()                                           # warm-up (allocator, oneDNN)
        for n in cands:
            torch.set_num_threads(n)
            sweep[n] = cpu_pass()
            if sweep[n] > 1.5 * min(sweep.values()):
                break
        cores = min(sweep, key=sweep.get)
        torch.set_num_threads(cores)
        n = max(1, min(5, int(12.0 / max(sweep[cores], 1e-3))))
        cpu_s = min(sweep[cores], sum(cpu_pass() for _ in range(n)) / n)
        out["cpu_baseline"] = {
            "value": B / cpu_s, "unit": "images/s", "cores": cores, "kind": "port",
            "all_cores": ({"cores": avail, "value": B / sweep[avail]} if avail in sweep else
                          "not reached: the sweep stopped at %d threads, already %.1f x slower "
                          "than the best setting" % (max(sweep), max(sweep.values()) / sweep[cores])),
            "thread_sweep_s_per_pass": {str(k): v for k, v in sweep.items()},
            "sample": "the same batch (%d image(s), same weights) through the CPU oracle%s, "
                      "torch CPU fp32: one warm-up pass, one pass at each of %s threads, then "
                      "%d timed pass(es) at the best setting (%d threads; host exposes %d)"
                      % (B, " (oracle backbone + head, i.e. the same image -> triplets path)"
                         if obb is not None else " head (simple_test_bboxes)", sorted(sweep), n,
                         cores, avail)}

    # The JSON line must be the LAST line of stdout: RCCL writes a version banner through C
    # stdio when NCCL_DEBUG=VERSION (set on these boxes), which -- buffered on a pipe -- would
    # otherwise land behind it at exit.  Every rank flushes C stdio, then the last barrier,
    # the communicator is torn down, C stdio is flushed once more, and only then rank 0 prints.
    def flush_c_stdio():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # pragma: no cover
            pass
    sys.stdout.flush()
    flush_c_stdio()
    if world > 1 or one_rank:
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
