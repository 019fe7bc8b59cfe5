"""CPU: the multi-GPU path (image sharding + triplet all-gather) with world_size 2
over gloo."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pairnet_amd.dist import (TripletGatherer, all_gather_triplets, pack_triplets,
                              shard_indices, triplet_record_len, unpack_triplets)


def _record(i, R=100, C=56):
    g = torch.Generator().manual_seed(1000 + i)
    labels = torch.randint(1, 134, (2 * R,), generator=g)
    rel = torch.rand(R, C + 1, generator=g)
    sub = torch.randint(0, 100, (R,), generator=g)
    obj = torch.randint(0, 100, (R,), generator=g)
    return pack_triplets(labels, rel, sub, obj)


def _worker(rank, world, port, n_images, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_indices(n_images, rank, world)
    local = torch.stack([_record(i) for i in mine]) if mine else \
        torch.zeros(0, triplet_record_len(100, 56))
    out = all_gather_triplets(local, n_images)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_indices_cover_dataset_like_distributed_sampler():
    for n, w in ((16, 8), (5, 2), (3, 4)):
        seen = sorted(i for r in range(w) for i in shard_indices(n, r, w))
        assert seen == list(range(n))
        assert shard_indices(n, 1, w) == list(range(1, n, w))


def test_pack_roundtrip():
    rec = _record(3)
    assert rec.shape[0] == triplet_record_len(100, 56)
    d = unpack_triplets(rec, 100, 56)
    assert d["labels"].dtype == torch.int64 and d["rel_dists"].shape == (100, 57)
    assert torch.equal(pack_triplets(d["labels"], d["rel_dists"], d["sub_pos"], d["obj_pos"]), rec)


def test_all_gather_triplets_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_images, world, port = 5, 2, _free_port()   # uneven split: rank 0 gets 3, rank 1 gets 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_images, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = torch.stack([_record(i) for i in range(n_images)])
    for r in range(world):
        assert torch.equal(outs[r], expect)


def _gatherer_worker(rank, world, port, n_local, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gt = TripletGatherer(n_local, 100, 56, "cpu")
    outs = []
    for step in range(2):                      # the same buffers serve every step
        for j in range(n_local):               # rank r holds images r, r + W, ...
            gt.send[j].copy_(_record(100 * step + rank + j * world))
        outs.append(gt.gather().clone())
    q.put((rank, [o.numpy() for o in outs], gt.records_gathered))   # (by value)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_local", [1, 2])
def test_preallocated_triplet_gatherer_world2_gloo(n_local):
    """The bench's per-step collective: one all-gather into preallocated buffers, records
    back in dataset order on every rank (rank r's j-th image is r + j*W)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port = 2, _free_port()
    procs = [ctx.Process(target=_gatherer_worker, args=(r, world, port, n_local, q))
             for r in range(world)]
    for p in procs:
        p.start()
    outs = {r: (o, n) for r, o, n in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        got, n = outs[r]
        assert n == 2 * world * n_local
        for step in range(2):
            expect = torch.stack([_record(100 * step + i) for i in range(world * n_local)])
            assert torch.equal(torch.from_numpy(got[step]), expect)


def _ring_worker(rank, world, port, n_local, steps, delay, q):
    """The pipelined bench's form: a ring of send buffers, the collective of the step `delay`
    steps back, flush() at the end."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gt = TripletGatherer(n_local, 100, 56, "cpu", ring=delay + 2)
    outs = []
    for step in range(steps):
        gt.begin_step()
        for j in range(n_local):
            gt.send[j].copy_(_record(100 * step + rank + j * world))
        gt.end_step()
        got = gt.gather_delayed(delay)
        if got is not None:
            outs.append(got.clone())
    outs += [o.clone() for o in gt.flush()]
    with pytest.raises(RuntimeError):          # (a full ring refuses another step)
        for _ in range(delay + 3):
            gt.begin_step()
            gt.end_step()
    q.put((rank, [o.numpy() for o in outs], gt.records_gathered))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_local,delay", [(1, 3), (2, 1)])
def test_delayed_ring_gatherer_world2_gloo(n_local, delay):
    """Every step's records come back exactly once, in step order and dataset order, whether
    they were gathered `delay` steps late or by flush()."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port, steps = 2, _free_port(), 6
    procs = [ctx.Process(target=_ring_worker, args=(r, world, port, n_local, steps, delay, q))
             for r in range(world)]
    for p in procs:
        p.start()
    outs = {r: (o, n) for r, o, n in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        got, n = outs[r]
        assert len(got) == steps and n == steps * world * n_local
        for step in range(steps):
            expect = torch.stack([_record(100 * step + i) for i in range(world * n_local)])
            assert torch.equal(torch.from_numpy(got[step]), expect)


class _HostHead:
    """What `multi_gpu_test` reads of a head."""
    num_rel_query, num_relations, device = 100, 56, None


class _HostDetector:
    """A detector stand-in for the CPU tests of the distributed LOOP (the real one needs the
    MI355X): `stream_triplets` yields, `lag` batches late like the pipeline, host tensors that
    are a deterministic function of the image; `released` counts the release hooks."""

    def __init__(self, lag=3):
        self.bbox_head, self.lag, self.released = _HostHead(), lag, 0

    @staticmethod
    def detect(img):
        i = int(img.flatten()[0])
        d = unpack_triplets(_record(i), 100, 56)
        res = (None, d["labels"], None, None, None, None, None, d["rel_dists"])
        return res, d["sub_pos"], d["obj_pos"]

    def stream_triplets(self, batches, rescale=False, depth=4):
        from pairnet_amd.dist import TripletBatch
        queue = []

        def rel(stream):
            self.released += 1
        for img, metas in batches:
            assert img.shape[0] == len(metas)
            got = [self.detect(img[b]) for b in range(img.shape[0])]      # (one record per image)
            queue.append(TripletBatch([g[0] for g in got], [g[1] for g in got],
                                      [g[2] for g in got], stream=None, release=rel))
            if len(queue) > self.lag:
                yield queue.pop(0)
        while queue:
            yield queue.pop(0)


class _ListEvaluator:
    """Stands in for TripletEvaluator: the "match lists" are a function of labels and GT."""

    def __call__(self, res, gt_rels, gt_labels, gt_masks):
        n = len(gt_rels)
        base = int(res[1][0])
        p2g = [[(base + r) % n] if n and r % 3 == 0 else [] for r in range(100)]
        rec = {k: len({g for l in p2g[:k] for g in l}) / float(n) for k in (20, 50, 100)} if n else None
        return dict(pred_to_gt=p2g, phrdet_pred_to_gt=p2g, sgdet_recall=rec, phrdet_recall=rec)


def _host_dataset(n):
    return [(torch.full((1, 3, 2, 2), float(i)), [dict(img_shape=(2, 2, 3), scale_factor=[1.0] * 4)])
            for i in range(n)]


def _host_annotations(n):
    import numpy as np
    out = []
    for i in range(n):
        g = 0 if i == 1 else 2 + i % 3       # (image 1 has no ground-truth relations: skipped)
        rels = np.array([[j % 2, (j + 1) % 2, 1 + (i + j) % 56] for j in range(g)]).reshape(-1, 3)
        out.append(dict(gt_rels=rels, gt_labels=np.array([3, 7]), gt_masks=None))
    return out


def _loop_worker(rank, world, port, n_images, q, k=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pairnet_amd.dist import multi_gpu_test
    from pairnet_amd.evaluation import SceneGraphMetrics
    det = _HostDetector()
    out = multi_gpu_test(det, _host_dataset(n_images), annotations=_host_annotations(n_images),
                         evaluator=_ListEvaluator(), metrics=SceneGraphMetrics(56), depth=2,
                         samples_per_gpu=k)
    q.put((rank, out["records"].numpy(), out["collectives"], det.released, out.get("metrics")))
    dist.barrier()
    dist.destroy_process_group()


def test_multi_gpu_test_world2_equals_world1_gloo():
    """The product loop end to end on the host: shard -> detect -> pack -> delayed ring
    all-gather -> dataset-order records on EVERY rank, evaluated where the images are, metrics
    on rank 0 -- with an uneven split (5 images over 2 ranks: rank 1 pads a zero step) the
    records and the metrics equal those of the one-process run."""
    from pairnet_amd.dist import multi_gpu_test
    from pairnet_amd.evaluation import SceneGraphMetrics
    n = 5
    det = _HostDetector()
    one = multi_gpu_test(det, _host_dataset(n), annotations=_host_annotations(n),
                         evaluator=_ListEvaluator(), metrics=SceneGraphMetrics(56), depth=2)
    assert one["world_size"] == 1 and one["collectives"] == n and det.released == n
    expect = torch.stack([_record(i) for i in range(n)])
    assert torch.equal(one["records"], expect)
    assert one["metrics"]["images"] == n - 1 and one["metrics"]["skipped"] == 1

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port = 2, _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = {r: rest for r, *rest in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        rec, ncoll, released, metrics = outs[r]
        assert torch.equal(torch.from_numpy(rec), expect)
        assert ncoll == 3                                  # ceil(5 / 2) collectives on BOTH ranks
        assert released == len(shard_indices(n, r, world))
    assert outs[1][3] is None and outs[0][3] == one["metrics"]


@pytest.mark.parametrize("n,k", [(5, 2), (7, 3), (4, 2)])
def test_multi_gpu_test_samples_per_gpu_world2_equals_world1_gloo(n, k):
    """BASELINE configs[2] through the product loop (tools/test.py:202-214: `samples_per_gpu`
    images per step): rank r batches its images r, r + W, ... k at a time; the records come
    back in dataset order on every rank, uneven tails (a last batch of fewer images, a rank
    with no batch left) are zero rows behind the dataset's end, and records and metrics equal
    those of the one-image-per-step, one-process run."""
    from pairnet_amd.dist import multi_gpu_test
    from pairnet_amd.evaluation import SceneGraphMetrics
    det = _HostDetector()
    one = multi_gpu_test(det, _host_dataset(n), annotations=_host_annotations(n),
                         evaluator=_ListEvaluator(), metrics=SceneGraphMetrics(56), depth=2)
    det = _HostDetector()
    onek = multi_gpu_test(det, _host_dataset(n), annotations=_host_annotations(n),
                          evaluator=_ListEvaluator(), metrics=SceneGraphMetrics(56), depth=2,
                          samples_per_gpu=k)
    expect = torch.stack([_record(i) for i in range(n)])
    assert torch.equal(onek["records"], expect) and onek["metrics"] == one["metrics"]
    assert onek["collectives"] == -(-n // k) and det.released == -(-n // k)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port = 2, _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, world, port, n, q, k)) for r in range(world)]
    for p in procs:
        p.start()
    outs = {r: rest for r, *rest in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    steps = -(-(-(-n // world)) // k)
    for r in range(world):
        rec, ncoll, released, metrics = outs[r]
        assert torch.equal(torch.from_numpy(rec), expect)
        assert ncoll == steps                              # the same collectives on BOTH ranks
        assert released == -(-len(shard_indices(n, r, world)) // k)
    assert outs[1][3] is None and outs[0][3] == one["metrics"]


def test_multi_gpu_test_of_an_empty_dataset_returns_no_records():
    """N == 0 -> an empty [0, L] record tensor, not a TypeError."""
    from pairnet_amd.dist import multi_gpu_test
    out = multi_gpu_test(_HostDetector(), [], depth=2)
    assert out["records"].shape == (0, triplet_record_len(100, 56)) and out["collectives"] == 0


def test_collate_pads_unequal_images_like_mmcv():
    """`collate` (the loader's batching, tools/test.py:202-214): equal sizes are stacked,
    unequal ones zero-padded at the bottom / right to the batch maximum; metas stay per image."""
    from pairnet_amd.dist import collate
    a, b = torch.ones(1, 3, 4, 6), 2 * torch.ones(1, 3, 5, 3)
    ma, mb = [dict(img_shape=(4, 6, 3))], [dict(img_shape=(5, 3, 3))]
    img, metas = collate([(a, ma), (b, mb)])
    assert img.shape == (2, 3, 5, 6) and metas == ma + mb
    assert torch.equal(img[0, :, :4, :], a[0]) and float(img[0, :, 4:, :].abs().sum()) == 0
    assert torch.equal(img[1, :, :, :3], b[0]) and float(img[1, :, :, 3:].abs().sum()) == 0
    img, metas = collate([(a, ma), (a + 1, ma)])
    assert img.shape == (2, 3, 4, 6) and torch.equal(img[1], a[0] + 1)
    assert collate([(a, ma)])[0] is a


def test_collector_pads_partial_batches_and_keeps_step_order():
    """`TripletCollector` on the host (no process group): two images per step, the last step
    with one image only (its second row is zeros), records kept in step order; the gather of a
    step is issued `depth` steps late and `finish()` collects the rest."""
    from pairnet_amd.dist import TripletBatch, TripletCollector
    col = TripletCollector(_HostHead(), depth=2, n_local=2, keep_steps=3)
    for step, ids in enumerate(((0, 1), (2, 3), (4,))):
        got = [_HostDetector.detect(torch.full((1,), float(i))) for i in ids]
        col.add(TripletBatch([g[0] for g in got], [g[1] for g in got], [g[2] for g in got]))
        assert col.stored == max(0, step + 1 - 2)            # delayed by `depth` steps
    rec = col.finish()
    assert col.stored == 3 and rec.shape == (6, triplet_record_len(100, 56))
    for i in range(5):
        assert torch.equal(rec[i], _record(i))
    assert float(rec[5].abs().sum()) == 0.0                  # the padded row of the last step


def test_ring_entry_waits_for_the_collective_that_read_it():
    """the ring's host counters only say a collective was ENQUEUED.  On a device the
    entry carries a `sent` event behind its all-gather, and begin_step() makes the packing
    stream wait for it; on the host (this test) the bookkeeping must still cycle."""
    gt = TripletGatherer(1, 100, 56, "cpu", ring=3)
    assert gt.sent == [None, None, None]
    for step in range(7):
        gt.begin_step()
        gt.send[0].copy_(_record(step))
        gt.end_step()
        out = gt.gather_delayed(1)
        if step >= 1:
            assert torch.equal(out[0], _record(step - 1))
    assert torch.equal(gt.flush()[0][0], _record(6))


@pytest.mark.gpu
def test_a_late_collective_does_not_hold_the_producing_stream():
    """a slow peer shows up on a rank as a collective that COMPLETES late.
    Injected here as a ~50 ms spin on the collector's side stream in front of one step's
    collective: the stream that produces and packs the records must run on undisturbed -- this
    step and the next three finish on it while the side stream is still held (its only tie to
    the side stream is a ring entry's `sent` event: step j re-packs the entry whose collective
    was issued at step j - ring + depth, ring = depth + 4) -- and every record still arrives,
    in order."""
    from pairnet_amd.dist import TripletBatch, TripletCollector

    class Head(_HostHead):
        device = torch.device("cuda:0")
    dev, depth, steps = Head.device, 2, 12
    col = TripletCollector(Head(), depth=depth, n_local=1, keep_steps=steps)
    producer = torch.cuda.Stream(dev)
    recs = []
    for i in range(steps):
        res, sub, obj = _HostDetector.detect(torch.full((1,), float(i)))
        recs.append((tuple(t.to(dev) if t is not None else None for t in res),
                     sub.to(dev), obj.to(dev)))
    torch.cuda.synchronize()
    held_at = 3
    hold = torch.cuda.Event()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    torch.cuda._sleep(1000000)                # (the spin kernel's clock rate is measured)
    t1.record()
    t1.synchronize()
    spin_cycles = int(1000000 * 50.0 / max(t0.elapsed_time(t1), 1e-3))     # ~50 ms
    still_held = None
    for i, (res, sub, obj) in enumerate(recs):
        if i == held_at:
            with torch.cuda.stream(col.side):
                torch.cuda._sleep(spin_cycles)
                hold.record(col.side)
        col.add(TripletBatch([res], [sub], [obj], stream=producer))
        if i == held_at + 3:
            # everything queued on the producing stream so far is done ...
            producer.synchronize()
            # ... while the side stream is still inside the injected delay
            still_held = not hold.query()
    rec = col.finish()
    assert still_held is True, "the producing stream waited for the held side stream"
    assert col.stored == steps
    for i in range(steps):
        assert torch.equal(rec[i].cpu(), _record(i)), i


@pytest.mark.gpu
def test_pack_triplets_kernel_equals_the_torch_packing():
    from pairnet_amd import hip
    g = torch.Generator().manual_seed(3)
    labels = torch.randint(1, 134, (200,), generator=g)
    rel = torch.rand(100, 57, generator=g)
    sub, obj = torch.randint(0, 100, (100,), generator=g), torch.randint(0, 100, (100,), generator=g)
    rec = torch.empty(triplet_record_len(100, 56), device="cuda:0")
    hip.pack_triplets(labels.cuda(), rel.cuda(), sub.cuda(), obj.cuda(), rec, 100, 57)
    assert torch.equal(rec.cpu(), pack_triplets(labels, rel, sub, obj))


@pytest.mark.gpu
def test_bench_two_ranks_complete_and_report_whole_job_rate():
    """Plain `python bench.py --gpus 2` (no launcher: the script re-executes itself through
    torch.distributed.run; gloo so that two ranks can share the test box's single GPU):
    every rank must leave every collective, and rank 0 prints one JSON line with the
    whole-job rate and the number of gathered triplet records."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["PAIRNET_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6",
           "--warmup", "3", "--height", "256", "--width", "320"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 2 and rec["value"] > 0
    assert rec["scaling"] == "weak" and "roofline" in rec
    assert rec["config"]["path"] == "image" and rec["config"]["backbone"] == "ResNet-50"
    assert rec["triplet_records_gathered"] >= 2 * 6 and rec["dist_backend"] == "gloo"


@pytest.mark.gpu
def test_bench_bbox_head_two_ranks():
    """`python bench.py --head bbox --gpus 2` (the sibling path; same self-launch, gloo on one
    GPU): both ranks leave the barriers and the max-over-ranks reduce, one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["PAIRNET_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--head", "bbox", "--gpus", "2",
           "--steps", "5", "--warmup", "3", "--height", "320", "--width", "416"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 2 and rec["value"] > 0
    assert rec["config"]["head"] == "bbox" and rec["pipeline_check"].startswith("labels")


_RCCL_ONE_RANK = r"""
import os, sys, socket, torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pairnet_amd.dist import TripletGatherer, pack_triplets, triplet_record_len
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
gt = TripletGatherer(2, 100, 56, dev, force_collective=True)
g = torch.Generator().manual_seed(5)
want = []
for i in range(2):
    labels = torch.randint(1, 134, (200,), generator=g)
    rel = torch.rand(100, 57, generator=g)
    sub, obj = torch.randint(0, 100, (100,), generator=g), torch.randint(0, 100, (100,), generator=g)
    gt.pack(i, labels.to(dev), rel.to(dev), sub.to(dev), obj.to(dev))
    want.append(pack_triplets(labels, rel, sub, obj))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    out = gt.gather()                       # all_gather_into_tensor through RCCL
side.synchronize()
t = torch.tensor([2.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)    # the bench's timing reduction
dist.barrier()
ok = torch.equal(out.cpu(), torch.stack(want)) and float(t) == 2.5 and gt.records_gathered == 2
print("RCCL_ONE_RANK_OK" if ok else "RCCL_ONE_RANK_MISMATCH", dist.get_backend())
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_rccl_executes_the_bench_collectives_with_one_rank():
    """What a 1-GPU box can run of the RCCL path: communicator set-up (backend "nccl" = RCCL)
    and the bench's three collectives -- the all-gather of the packed triplet records on a side
    stream, the MAX reduction of the timing scalar, the barrier -- with world_size 1.  (Two
    ranks cannot share one GPU under RCCL; the multi-rank control flow is the gloo tests above.)
    Skipped, not failed, where RCCL cannot initialise at all."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    try:
        out = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, root], env=env, cwd=root,
                             capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL single-rank initialisation did not finish within 300 s on this box")
    if "RCCL_ONE_RANK_" not in out.stdout:
        pytest.skip("RCCL could not initialise here: " + out.stderr[-400:])
    assert "RCCL_ONE_RANK_OK nccl" in out.stdout, out.stdout[-400:]


@pytest.mark.gpu
def test_bench_json_is_the_last_stdout_line_with_rccl_in_the_loop():
    """`bench.py --rccl-one-rank`: the pipelined loop with a live RCCL communicator (its
    watchdog thread beside graph capture and replay), every step's records through
    all_gather_into_tensor, barrier + MAX reduction around the timed region -- and, with
    NCCL_DEBUG=VERSION as on the GPU boxes, RCCL's banner (written through C stdio, buffered
    on a pipe) must not land behind the JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["NCCL_DEBUG"] = "VERSION"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--rccl-one-rank", "--steps", "8",
           "--warmup", "3", "--height", "256", "--width", "320", "--no-extras",
           "--no-cpu-baseline"]
    try:
        out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=400)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL single-rank run did not finish within 400 s on this box")
    if out.returncode != 0 and "NCCL" in out.stderr.upper():
        pytest.skip("RCCL could not initialise here: " + out.stderr[-300:])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert lines[-1].startswith("{"), lines[-3:]
    rec = json.loads(lines[-1])
    assert rec["dist_backend"] == "nccl" and rec["rccl_ranks"] == 1 and rec["n_gpus"] == 1
    assert rec["triplet_records_gathered"] >= 8 and rec["value"] > 0
    assert rec["config"]["collective"].startswith("RCCL all-gather")
    assert sum(l.startswith("{") for l in lines) == 1


# ---- the product loop on the MI355X -------------------------------------------------------
_LOOP_SCRIPT = r"""
import os, sys, json, socket, torch
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
backend, rank, world, port, out_path = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
k = int(sys.argv[7]) if len(sys.argv) > 7 else 1          # samples_per_gpu
lazy = len(sys.argv) > 8 and sys.argv[8] == "lazy"       # annotations built on demand, on the device
from oracle.backbone import seeded_backbone_state      # (seeded weights only: test infrastructure)
from pairnet_amd import build_detector, pairnet_r50
from pairnet_amd.dist import multi_gpu_test
from pairnet_amd.evaluation import SceneGraphMetrics, TripletEvaluator
torch.cuda.set_device(0)
if world > 0:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group(backend, rank=rank, world_size=world)
det = build_detector(pairnet_r50())
det.backbone.load_state_dict(seeded_backbone_state(41))
det.bbox_head.init_weights(seed=3)
det.to("cuda:0")
H, W, N = 160, 224, 5
metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0] * 4)]
g = torch.Generator().manual_seed(3)
data = [(torch.randn(1, 3, H, W, generator=g).to("cuda:0"), metas) for _ in range(N)]
ga = torch.Generator().manual_seed(11)
ann = []
for i in range(N):
    gm = torch.rand(3, 80, 112, generator=ga) > 0.5
    rels = np.array([[0, 1, 5], [1, 2, 7], [2, 0, 9]]) if i != 2 else np.zeros((0, 3), dtype=np.int64)
    # (odd images: masks already on the device, as pairnet_amd.dataset.eval_ground_truth leaves them)
    ann.append(dict(gt_rels=rels, gt_labels=np.array([1 + i, 20, 90]),
                    gt_masks=gm.to("cuda:0") if i % 2 else gm.numpy()))
if lazy:
    class LazyAnn:
        # what pairnet_amd.dataset.eval_ground_truth does per image, WHEN the loop asks for it:
        # an H2D copy from pinned memory and a kernel on the caller's current stream -- the
        # evaluator reads the masks on the chain stream that produced the result (ADVICE r4)
        def __init__(self, anns):
            self.anns = anns
            self.pinned = [torch.as_tensor(a["gt_masks"]).cpu().pin_memory() for a in anns]
        def __len__(self):
            return len(self.anns)
        def __getitem__(self, i):
            m = self.pinned[i].to("cuda:0", non_blocking=True)
            big = torch.zeros(64, 1 << 20, device="cuda:0").cumsum(1)     # (keeps the stream busy)
            m = (m.to(torch.uint8) + (big[0, :1] * 0).to(torch.uint8)) > 0
            return dict(self.anns[i], gt_masks=m)
    ann = LazyAnn(ann)
if world == 0 and k == 1:      # the optional stream -> hardware-queue calibration of the detector's pipeline
    times = det.calibrate_pipeline(*data[0], depth=3, steps=2)
    assert len(times) == 4 and all(t > 0 for t in times) and det.pipeline(3) is det.pipeline(3)
    assert det.bbox_head.grid_reserve == 0 and not det.bbox_head.use_graphs
out = multi_gpu_test(det, data, annotations=ann, evaluator=TripletEvaluator(),
                     metrics=SceneGraphMetrics(56), depth=3,
                     force_collective=(backend == "nccl"), samples_per_gpu=k)
if rank == 0:
    # second call in the same process: the cached plans / graphs serve it
    again = multi_gpu_test(det, data, depth=3, force_collective=(backend == "nccl"),
                           samples_per_gpu=k) if world <= 1 else None
    np.savez(out_path, records=out["records"].cpu().numpy(), collectives=out["collectives"],
             metrics=json.dumps(out.get("metrics")),
             again=(again["records"].cpu().numpy() if again is not None else np.zeros(0)))
if world > 0:
    dist.barrier()
    dist.destroy_process_group()
print("LOOP_DONE", rank)
"""


def _run_loop(tmp_path, backend, world, *extra):
    import subprocess
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out_path = str(tmp_path / ("loop_%s_%d%s.npz" % (backend, world, "_".join(("",) + extra))))
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, "-c", _LOOP_SCRIPT, root, backend, str(r),
                               str(world), port, out_path] + list(extra), env=env, cwd=root,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(max(world, 1))]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q_ in procs:
                q_.kill()
            raise
    return procs, outs, (np.load(out_path) if os.path.exists(out_path) else None)


@pytest.mark.gpu
def test_multi_gpu_test_on_the_gpu_equals_simple_test_and_two_gloo_ranks(tmp_path):
    """`dist.multi_gpu_test` with the real detector: (a) one process, no process group: every
    image's record equals the one packed from `PSGTr.simple_test`-path outputs, the evaluator
    ran on the device results; (b) two ranks sharing this box's GPU over gloo (host-staged
    collective), uneven split: the same records in dataset order and the same metrics."""
    import json
    import numpy as np
    from oracle.backbone import seeded_backbone_state
    from pairnet_amd import build_detector, pairnet_r50
    procs, outs, one = _run_loop(tmp_path, "none", 0)
    assert procs[0].returncode == 0, outs[0][1][-3000:]
    det = build_detector(pairnet_r50())
    det.backbone.load_state_dict(seeded_backbone_state(41))
    det.bbox_head.init_weights(seed=3)
    det.to("cuda:0")
    H, W = 160, 224
    metas = [dict(img_shape=(H, W, 3), scale_factor=[2.0] * 4)]
    g = torch.Generator().manual_seed(3)
    head = det.bbox_head
    for i in range(5):
        img = torch.randn(1, 3, H, W, generator=g).to("cuda:0")
        res = head.simple_test(det.extract_feat(img), metas)[0]
        sub, obj = head.pair_positions()
        want = pack_triplets(res[1].cpu(), res[7].cpu(), sub[0].cpu(), obj[0].cpu())
        assert np.array_equal(one["records"][i], want.numpy()), i
    assert int(one["collectives"]) == 5 and np.array_equal(one["again"], one["records"])
    m1 = json.loads(str(one["metrics"]))
    assert m1["images"] == 4 and m1["skipped"] == 1 and "sgdet_mean_recall" in m1
    procs, outs, two = _run_loop(tmp_path, "gloo", 2)
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    assert np.array_equal(two["records"], one["records"]) and int(two["collectives"]) == 3
    assert json.loads(str(two["metrics"])) == m1


@pytest.mark.gpu
def test_multi_gpu_test_two_samples_per_gpu_and_lazy_device_annotations(tmp_path):
    """BASELINE configs[2] through product code (tools/test.py:202-214, `samples_per_gpu=2`):
    (a) one process: every record equals the one packed from `simple_test` on the same
    two-image batch; (b) two gloo ranks on this GPU, uneven split: the same records in dataset
    order and the same metrics; (c) ADVICE r4: annotations produced lazily ON THE DEVICE while
    the loop runs (H2D copy + kernels on the caller's stream) give the metrics of the eagerly
    loaded ones -- the evaluator's chain stream is ordered behind the stream that made them."""
    import json
    import numpy as np
    from oracle.backbone import seeded_backbone_state
    from pairnet_amd import build_detector, pairnet_r50
    procs, outs, one = _run_loop(tmp_path, "none", 0, "2")
    assert procs[0].returncode == 0, outs[0][1][-3000:]
    det = build_detector(pairnet_r50())
    det.backbone.load_state_dict(seeded_backbone_state(41))
    det.bbox_head.init_weights(seed=3)
    det.to("cuda:0")
    H, W = 160, 224
    meta = dict(img_shape=(H, W, 3), scale_factor=[2.0] * 4)
    g = torch.Generator().manual_seed(3)
    imgs = [torch.randn(1, 3, H, W, generator=g).to("cuda:0") for _ in range(5)]
    head = det.bbox_head
    for grp in ((0, 1), (2, 3), (4,)):
        res = head.simple_test(det.extract_feat(torch.cat([imgs[i] for i in grp])), [meta] * len(grp))
        sub, obj = head.pair_positions()
        for j, i in enumerate(grp):
            want = pack_triplets(res[j][1].cpu(), res[j][7].cpu(), sub[j].cpu(), obj[j].cpu())
            assert np.array_equal(one["records"][i], want.numpy()), i
    assert int(one["collectives"]) == 3 and np.array_equal(one["again"], one["records"])
    m1 = json.loads(str(one["metrics"]))
    assert m1["images"] == 4 and m1["skipped"] == 1
    # (b) rank 0 batches images (0, 2), (4,), rank 1 (1, 3): each image's record is the one
    # of its own two-image batch
    procs, outs, two = _run_loop(tmp_path, "gloo", 2, "2")
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    assert int(two["collectives"]) == 2 and two["records"].shape == one["records"].shape
    for grp in ((0, 2), (1, 3), (4,)):
        res = head.simple_test(det.extract_feat(torch.cat([imgs[i] for i in grp])), [meta] * len(grp))
        sub, obj = head.pair_positions()
        for j, i in enumerate(grp):
            want = pack_triplets(res[j][1].cpu(), res[j][7].cpu(), sub[j].cpu(), obj[j].cpu())
            assert np.array_equal(two["records"][i], want.numpy()), i
    # (c) lazily built device annotations
    procs, outs, lazy = _run_loop(tmp_path, "none", 0, "2", "lazy")
    assert procs[0].returncode == 0, outs[0][1][-3000:]
    assert np.array_equal(lazy["records"], one["records"])
    assert json.loads(str(lazy["metrics"])) == m1


@pytest.mark.gpu
def test_multi_gpu_test_through_rccl_with_one_rank(tmp_path):
    """The same loop with a live RCCL communicator (world_size 1, `force_collective`): every
    step's records go through all_gather_into_tensor on the side stream."""
    import numpy as np
    try:
        procs, outs, got = _run_loop(tmp_path, "nccl", 1)
    except Exception as e:           # (timeout)
        pytest.skip("RCCL single-rank run did not finish: %r" % (e,))
    if procs[0].returncode != 0 and "NCCL" in outs[0][1].upper():
        pytest.skip("RCCL could not initialise here: " + outs[0][1][-300:])
    assert procs[0].returncode == 0, outs[0][1][-3000:]
    _, _, one = _run_loop(tmp_path, "none", 0)
    assert np.array_equal(got["records"], one["records"]) and int(got["collectives"]) == 5


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["baseline_r50", "psgtr2_r50", "cross_r101_vg"])
def test_multi_gpu_test_with_the_sibling_heads(model):
    """The product loop over the sibling detectors (BASELINE configs[4]: same loop, other
    heads): CrossHeadBaseline (matched object-query rows), PSGTrHead2 (query i is triplet i),
    both through the pipeline, and the box-trunk CrossHeadBBox (neck: one batch at a time) --
    every record equals the one packed from the head's own synchronous outputs."""
    import pairnet_amd as P
    from pairnet_amd.dist import multi_gpu_test, unpack_triplets
    cfg = getattr(P, model)()
    det = P.build_detector(cfg.model if "model" in cfg else cfg)
    det.bbox_head.init_weights(seed=5)
    det.to("cuda:0")
    g = torch.Generator().manual_seed(21)
    # (round 5: three image sizes in one pass -- every head's plans, the box trunk's and its
    # neck's included, are views of per-slot arenas; sizes come back after others have passed)
    sizes = [(160, 224), (160, 224), (192, 160), (128, 256), (160, 224), (192, 160)]
    data = [(torch.randn(1, 3, H, W, generator=g).to("cuda:0"),
             [dict(img_shape=(H, W, 3), scale_factor=[2.0] * 4, batch_input_shape=(H, W))])
            for H, W in sizes]
    head = det.bbox_head
    want = []
    for img, m in data:
        res = head.simple_test(det.extract_feat(img), m)[0]
        sub, obj = head.pair_positions()
        want.append(pack_triplets(res[1].cpu(), res[-1].cpu(), sub[0].cpu(), obj[0].cpu()))
    out = multi_gpu_test(det, data, depth=3)
    assert out["records"].shape == (len(data), triplet_record_len(head.num_rel_query, head.num_relations))
    for i in range(len(data)):
        assert torch.equal(out["records"][i].cpu(), want[i]), (model, i)
    d = unpack_triplets(out["records"][0].cpu(), head.num_rel_query, head.num_relations)
    assert d["rel_dists"].shape == (head.num_rel_query, head.num_relations + 1)


def _reducer_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pairnet_amd.dist import GradReducer
    n = 5000
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = GradReducer(flat, bucket_bytes=4096)            # 1024 floats per bucket: 5 buckets
    log = []
    for _ in range(2):                                    # two "backward passes"
        flat.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
        red.start()
        for end in (100, 1024, 3000, 3072):               # the backward pass's ready() calls
            red.ready(end)
            log.append(red.next)
        red.finish()
        log.append(red.next)
    q.put((rank, flat.clone(), red.scale, red.collectives, log, [b for b in red.bounds]))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_buckets_world2_gloo():
    """`GradReducer` (the training step's DDP half) over gloo: buckets are all-reduced as soon as
    the backward pass reports a prefix complete, never before; after finish() every rank holds
    the SUM, and `scale` = 1 / world is what the optimizer kernels multiply by."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = torch.arange(5000, dtype=torch.float32) * 3.0          # rank 0: x1, rank 1: x2
    for rank, flat, scale, ncoll, log, bounds in got:
        assert torch.equal(flat, want), rank
        assert scale == 0.5 and ncoll == 10
        assert bounds == [(0, 1024), (1024, 2048), (2048, 3072), (3072, 4096), (4096, 5000)]
        # buckets issued after ready(100), ready(1024), ready(3000), ready(3072), finish()
        assert log == [0, 1, 2, 3, 5] * 2


def test_grad_reducer_is_a_no_op_without_a_process_group():
    from pairnet_amd.dist import GradReducer
    flat = torch.ones(300)
    red = GradReducer(flat)
    red.start()
    red.ready(300)
    red.finish()
    assert red.scale == 1.0 and red.collectives == 0 and torch.equal(flat, torch.ones(300))


def test_numa_binding_reads_sysfs_and_is_best_effort(tmp_path):
    """`dist.bind_to_gpu_numa` (bench.py --gpus N binds every rank to its GPU's NUMA node):
    the node and cpu list come from sysfs; unknown topology (node -1, no files, no GPU) means
    "leave the affinity alone", never an error."""
    from pairnet_amd.dist import _cpulist, bind_to_gpu_numa, gpu_numa_cpus
    assert _cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and _cpulist("") == []
    root = tmp_path / "sys"
    dev = root / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = root / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("64-127\n")
    assert gpu_numa_cpus("0000:C1:00.0", str(root)) == (1, list(range(64, 128)))
    (dev / "numa_node").write_text("-1\n")
    assert gpu_numa_cpus("0000:c1:00.0", str(root)) == (None, None)
    assert gpu_numa_cpus("0000:ff:00.0", str(root)) == (None, None)
    if not torch.cuda.is_available():
        assert bind_to_gpu_numa(0) is None       # no GPU: nothing is touched
