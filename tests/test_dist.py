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
