"""Multi-GPU: shard images over ranks, all-gather the predicted triplets.

The reference's only parallelism is data parallel inference: `DistributedSampler(
shuffle=False)` hands rank r the images r, r+W, ... and mmdet's
`collect_results_gpu` pickles whole `Result` objects (masks included, ~61 MB/img)
through two NCCL all_gathers (tools/test.py:256-267).
Here images are independent too (no cross-image op in pairnet_head.py:260-417), so
ranks never talk on the data path; the single collective gathers a compact
fixed-shape triplet record per image (~27 KB) with one RCCL all-gather over xGMI
(`torch.distributed` backend "nccl" on ROCm; "gloo" in the CPU tests).  Masks and
panoptic maps stay on the producing GPU.
"""
import torch
import torch.distributed as dist


def shard_indices(num_images, rank, world_size):
    """Indices of the images rank `rank` processes (DistributedSampler(shuffle=False)
    without padding): rank, rank + W, rank + 2W, ..."""
    return list(range(rank, num_images, world_size))


def _cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_cpus(pci_bdf, sysfs="/sys"):
    """(numa node, cpu list) of the PCI device `pci_bdf` ("0000:c1:00.0") from sysfs, or
    (None, None) when the platform does not say (node -1, missing files)."""
    try:
        with open("%s/bus/pci/devices/%s/numa_node" % (sysfs, pci_bdf.lower())) as f:
            node = int(f.read().strip())
        if node < 0:
            return None, None
        with open("%s/devices/system/node/node%d/cpulist" % (sysfs, node)) as f:
            return node, _cpulist(f.read())
    except (OSError, ValueError):
        return None, None


def bind_to_gpu_numa(device_index, sysfs="/sys"):
    """One process per GPU (tools/dist_test.sh:7 launches them unpinned): keep this rank's host
    threads on the NUMA node its GPU hangs off, so that the ~330 kernel launches per image (or
    the graph replays) and the pinned-memory traffic of 8 ranks do not cross sockets.  Best
    effort: returns dict(numa_node, cpus) on success, None when the topology is unknown or the
    affinity cannot be set (containers with a restricted cpuset keep what they have)."""
    import os
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception:       # noqa: BLE001 -- no GPU, or a torch without the PCI fields
        return None
    node, cpus = gpu_numa_cpus(bdf, sysfs)
    if not cpus or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = os.sched_getaffinity(0)
    want = sorted(set(cpus) & allowed)
    if not want:
        return None
    try:
        os.sched_setaffinity(0, want)
    except OSError:
        return None
    return dict(numa_node=node, cpus=len(want), pci=bdf)


def triplet_record_len(num_rel_query, num_relations):
    """float32 words per image: labels(2R) | rel_dists(R*(C+1)) | sub_pos(R) | obj_pos(R)."""
    return 2 * num_rel_query + num_rel_query * (num_relations + 1) + 2 * num_rel_query


def pack_triplets(labels, rel_dists, sub_pos, obj_pos):
    """Per-image tensors -> one float32 record (indices < 2^24 are exact in fp32)."""
    return torch.cat([labels.to(torch.float32).flatten(), rel_dists.flatten().to(torch.float32),
                      sub_pos.to(torch.float32).flatten(), obj_pos.to(torch.float32).flatten()])


def unpack_triplets(rec, num_rel_query, num_relations):
    R, C = num_rel_query, num_relations + 1
    o = 0
    labels = rec[o:o + 2 * R].to(torch.int64); o += 2 * R
    rel_dists = rec[o:o + R * C].view(R, C); o += R * C
    sub_pos = rec[o:o + R].to(torch.int64); o += R
    obj_pos = rec[o:o + R].to(torch.int64)
    return dict(labels=labels, rel_dists=rel_dists, sub_pos=sub_pos, obj_pos=obj_pos,
                rel_pairs=torch.arange(2 * R, dtype=torch.int32).reshape(2, -1).T)


def all_gather_triplets(local_records, num_images, group=None):
    """local_records: [n_local, L] float32 (rows in the order of shard_indices).
    Returns [num_images, L] in dataset order on every rank.  Ranks with fewer images
    pad to the per-rank maximum, like collect_results_gpu pads to the longest pickle."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_records[:num_images]
    W, rank = dist.get_world_size(group), dist.get_rank(group)
    per_rank = (num_images + W - 1) // W
    L = local_records.shape[1]
    send = local_records.new_zeros((per_rank, L))
    send[:local_records.shape[0]] = local_records
    recv = local_records.new_empty((W * per_rank, L))
    try:
        dist.all_gather_into_tensor(recv, send, group=group)
    except (RuntimeError, NotImplementedError):  # older gloo: list form
        parts = [torch.empty_like(send) for _ in range(W)]
        dist.all_gather(parts, send, group=group)
        recv = torch.cat(parts, 0)
    # rank r's j-th row is image r + j*W: interleave (zip over ranks) and truncate
    out = recv.view(W, per_rank, L).transpose(0, 1).reshape(W * per_rank, L)
    return out[:num_images].contiguous()


class TripletGatherer:
    """The per-step collective of a data-parallel run with everything preallocated:
    `pack(i, result, sub_pos, obj_pos)` writes image i's record straight into the send
    buffer (one small HIP kernel, csrc/postproc.hip k_pack_triplets; no torch.cat), and
    `gather()` issues ONE all-gather (RCCL over xGMI for backend "nccl") into a
    preallocated receive buffer and returns the records in dataset order
    ([world * n_local, L], a view of an internal buffer).

    `ring` > 1 (the pipelined bench): `ring` send buffers used in turn, so that a step's
    records can be packed on the stream that produced them while the collective of an OLDER
    step -- `gather_delayed(delay)`, ordered behind that step's `packed` event, which by then
    has long fired -- runs on a side stream.  Why: a command on a side stream that has to WAIT
    (for the newest query chain to finish) blocks whatever else HIP has mapped onto the same
    hardware queue, a stage-A stream of the pipeline as likely as not: measured 188 instead of
    203 images/s per GPU with the pack + collective of the newest result on a side stream."""

    def __init__(self, n_local, num_rel_query, num_relations, device, group=None,
                 force_collective=False, ring=1):
        from . import hip
        self.hip, self.group = hip, group
        self.R, self.C1 = num_rel_query, num_relations + 1
        self.L = triplet_record_len(num_rel_query, num_relations)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.n_local = n_local
        self.on_device = torch.device(device).type == "cuda"
        self.ring = max(1, int(ring))
        self.sends = [torch.zeros((n_local, self.L), device=device, dtype=torch.float32)
                      for _ in range(self.ring)]
        self.send = self.sends[0]
        self.recv = torch.empty((self.world * n_local, self.L), device=device, dtype=torch.float32)
        self.out = torch.empty_like(self.recv)
        self.records_gathered = 0
        # run the collective even with one rank (what a 1-GPU box can execute of the RCCL
        # path: communicator set-up and the all-gather itself, tests/test_dist.py)
        self.force_collective = bool(force_collective) and dist.is_initialized()
        self.packed = [None] * self.ring      # per ring entry: event behind its pack kernels
        # per ring entry: event behind the all-gather that READ it (recorded on the stream the
        # collective was issued on).  The host counters below only say that the collective has
        # been ENQUEUED; a rank whose collectives are backed up behind a slow peer would
        # otherwise re-pack an entry RCCL has not read yet (write-after-read).
        self.sent = [None] * self.ring
        self.filled = 0                       # steps packed so far
        self.gathered = 0                     # steps gathered so far

    # ---- ring form -----------------------------------------------------------------------
    def begin_step(self):
        """Select the ring entry of a new step: pack() then writes there (on the caller's
        current stream); finish with end_step()."""
        if self.filled - self.gathered >= self.ring:
            raise RuntimeError("TripletGatherer ring is full: gather_delayed() / flush() first")
        k = self.filled % self.ring
        if self.on_device and self.sent[k] is not None:
            # the packing stream may not overwrite the entry before the collective that read
            # it has finished (a device-side wait: the host does not block)
            torch.cuda.current_stream().wait_event(self.sent[k])
        self.send = self.sends[k]

    def end_step(self):
        if self.on_device:
            ev = self.packed[self.filled % self.ring] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.packed[self.filled % self.ring] = ev
        self.filled += 1

    def gather_delayed(self, delay, host_staging=False, consume=None):
        """All-gather the oldest packed step if it is at least `delay` steps old (on the
        caller's current stream, behind that step's `packed` event); returns its records or
        None.  `consume(records)`: reader of the returned buffer, run BEFORE the ring entry is
        marked as read -- with one rank the records ARE the ring entry, so a reader that ran
        after this call would not be ordered in front of the entry's next pack."""
        if self.filled - self.gathered <= delay:
            return None
        k = self.gathered % self.ring
        if self.on_device and self.packed[k] is not None:
            torch.cuda.current_stream().wait_event(self.packed[k])
        self.send = self.sends[k]
        out = self.gather(host_staging=host_staging)
        if consume is not None:
            out = consume(out)
        if self.on_device:
            ev = self.sent[k] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.sent[k] = ev
        self.gathered += 1
        return out

    def flush(self, host_staging=False):
        """Gather every step still in the ring (end of the run)."""
        outs = []
        while self.gathered < self.filled:      # (copies: gather() returns an internal buffer)
            outs.append(self.gather_delayed(0, host_staging=host_staging,
                                            consume=lambda o: o.clone()))
        return outs

    def pack(self, i, labels, rel_dists, sub_pos, obj_pos):
        if not self.on_device:     # host records (the gloo tests of the control flow)
            self.send[i].copy_(pack_triplets(labels, rel_dists, sub_pos, obj_pos))
            return
        self.hip.pack_triplets(labels, rel_dists, sub_pos, obj_pos, self.send[i], self.R, self.C1)

    def pack_padding(self, i):
        """A rank without an image for row i of this step still takes part in the collective:
        its row is zeros (dropped when the records are truncated to the dataset length)."""
        self.send[i].zero_()

    def gather(self, host_staging=False):
        """`host_staging`: run the collective on host copies (backend "gloo": the
        single-GPU functional check of the multi-rank control flow)."""
        if self.world == 1 and not self.force_collective:
            self.records_gathered += self.n_local
            return self.send
        send, recv = (self.send.cpu(), self.recv.cpu()) if host_staging else (self.send, self.recv)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        self.records_gathered += self.world * self.n_local
        if self.n_local == 1:          # one image per rank: rank order IS dataset order
            return recv
        # rank r's j-th row is image r + j*W: interleave (zip over ranks)
        out = self.out.cpu() if host_staging else self.out
        out.view(self.n_local, self.world, self.L).copy_(
            recv.view(self.world, self.n_local, self.L).transpose(0, 1))
        return out


class TripletBatch:
    """One batch of `get_bboxes` tuples as a detector's `stream_triplets()` yields them:
    `results[i]` is image i's tuple (labels at [1], r_dists last), `sub_pos[i]` / `obj_pos[i]`
    the query rows of its R triplets, `stream` the stream they were produced on (None on the
    host) and `release(stream)` tells the producer that its buffers have been read on `stream`
    up to this point."""
    __slots__ = ("results", "sub_pos", "obj_pos", "stream", "release")

    def __init__(self, results, sub_pos, obj_pos, stream=None, release=None):
        self.results, self.sub_pos, self.obj_pos = results, sub_pos, obj_pos
        self.stream, self.release = stream, release


class TripletCollector:
    """Pack-and-gather of a data-parallel run, the part of mmdet's `collect_results_*`
    (tools/test.py:262-267) that runs per step: `add(batch)` packs a `TripletBatch`'s records
    into a `TripletGatherer` ring entry ON THE STREAM THAT PRODUCED THEM, releases the
    producer's buffers, and issues the all-gather of the step `depth` steps back on a side
    stream (its inputs have long been written: the collective never makes a hardware queue
    wait, and a slow peer never stalls a compute stream).  `pad()` contributes a zero step
    (a rank that has run out of images still takes part in every collective); `finish()`
    gathers what is left in the ring and, with `keep_steps`, returns every step's records
    [steps * world * n_local, L] in dataset order (image (s * n_local + j) * W + r is row j
    of rank r in step s)."""

    def __init__(self, head, depth=4, n_local=1, group=None, host_staging=None,
                 force_collective=False, keep_steps=0):
        ini = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if ini else 1
        self.device = torch.device(getattr(head, "device", None) or "cpu")
        self.on_gpu = self.device.type == "cuda"
        if host_staging is None:    # gloo cannot take device tensors: stage through the host
            host_staging = self.on_gpu and ini and dist.get_backend(group) != "nccl"
        self.host_staging, self.depth, self.n_local = bool(host_staging), int(depth), int(n_local)
        self.gatherer = TripletGatherer(n_local, head.num_rel_query, head.num_relations,
                                        self.device, group=group,
                                        force_collective=force_collective, ring=depth + 4)
        self.side = torch.cuda.Stream(self.device) if self.on_gpu else None
        self.rows = self.world * self.n_local
        store_dev = torch.device("cpu") if self.host_staging else self.device
        self.records = torch.zeros((keep_steps * self.rows, self.gatherer.L), device=store_dev,
                                   dtype=torch.float32) if keep_steps else None
        self.stored = 0
        if self.on_gpu:
            # the buffers above were zero-filled on the CURRENT stream; they are written next
            # on other streams (chain streams, the side stream): finish the fills first
            torch.cuda.current_stream(self.device).synchronize()

    def _on(self, stream):
        import contextlib
        if self.on_gpu and stream is not None:
            return torch.cuda.stream(stream)
        return contextlib.nullcontext()

    def _store(self, out):
        if self.records is not None:
            k = self.stored      # (gather() hands out an internal buffer: keep a copy)
            self.records[k * self.rows:(k + 1) * self.rows].copy_(out, non_blocking=True)
            self.stored = k + 1
        return out

    def _collect(self, delay):
        with self._on(self.side):
            self.gatherer.gather_delayed(delay, host_staging=self.host_staging,
                                         consume=self._store)

    def add(self, tb, before_release=None):
        """`tb`: a TripletBatch of `n_local` images.  `before_release(tb)`: optional reader of
        the batch's device tensors, run on the producing stream before they are released."""
        g = self.gatherer
        ps = tb.stream if tb.stream is not None else (
            torch.cuda.current_stream(self.device) if self.on_gpu else None)
        with self._on(ps):
            g.begin_step()
            for i, res in enumerate(tb.results):
                g.pack(i, res[1], res[-1], tb.sub_pos[i], tb.obj_pos[i])
            for i in range(len(tb.results), self.n_local):
                g.pack_padding(i)
            g.end_step()
            if before_release is not None:
                before_release(tb)
            if tb.release is not None:    # the producer's slot may be reused once `ps` is here
                tb.release(ps)
        self._collect(self.depth)

    def pad(self):
        g = self.gatherer
        g.begin_step()
        for i in range(self.n_local):
            g.pack_padding(i)
        g.end_step()
        self._collect(self.depth)

    def finish(self):
        g = self.gatherer
        with self._on(self.side):
            while g.gathered < g.filled:
                g.gather_delayed(0, host_staging=self.host_staging, consume=self._store)
        if self.side is not None:
            self.side.synchronize()
        return self.records


def collate(items):
    """mmcv's `collate` for `samples_per_gpu` test images (tools/test.py:202-214 builds the
    loader with it): `items` = [(img [1, 3, H, W], [meta])], one image each -> (imgs
    [k, 3, Hmax, Wmax], [meta] * k).  Images of different sizes are zero-padded at the bottom /
    right to the largest of the batch, as the DataContainer stacking does; every image keeps
    its own `img_metas` entry (`img_shape`, `scale_factor`)."""
    if len(items) == 1:
        return items[0]
    if str(getattr(items[0][0], "dtype", "")).endswith("uint8"):
        # DECODED images (uint8 (H, W, 3)): the detector's own test pipeline resizes, normalises
        # and pads them into one batch tensor on the GPU (`TestPipeline.batch`)
        return [it[0] for it in items], None
    imgs = [it[0] for it in items]
    metas = [m for it in items for m in it[1]]
    H, W = max(i.shape[-2] for i in imgs), max(i.shape[-1] for i in imgs)
    if all(tuple(i.shape[-2:]) == (H, W) for i in imgs):
        return torch.cat(imgs, 0), metas
    out = imgs[0].new_zeros((sum(i.shape[0] for i in imgs), imgs[0].shape[1], H, W))
    row = 0
    for i in imgs:
        out[row:row + i.shape[0], :, :i.shape[-2], :i.shape[-1]] = i
        row += i.shape[0]
    return out, metas


def multi_gpu_test(detector, dataset, annotations=None, evaluator=None, metrics=None, *,
                   depth=4, group=None, host_staging=None, force_collective=False,
                   rescale=False, calibrate=True, samples_per_gpu=1):
    """The reference's distributed test loop -- mmdet `multi_gpu_test` + `collect_results_*`
    (tools/test.py:256-267) followed by `dataset.evaluate` (:277-295 ->
    pairnet/datasets/psg.py:285-404) -- for one process per GPU:

      * rank r takes the images r, r + W, ... (`shard_indices`: DistributedSampler(
        shuffle=False)), `samples_per_gpu` consecutive ones of them per step (the loader's
        batches, tools/test.py:202-214; `collate` pads unequal sizes like mmcv's), and runs
        them through `detector.stream_triplets` (PSGTr: backbone -> PipelinedHead, `depth`
        batches in flight);
      * every image's triplet record (labels | rel_dists | sub_pos | obj_pos, ~27 KB) is
        packed on the stream that produced it and all-gathered `depth` steps later on a side
        stream (`TripletGatherer` ring; RCCL for backend "nccl", host-staged for "gloo"), so
        neither the collective nor a slow peer stalls a compute stream; ranks that run out of
        images contribute zero rows, so every rank issues the same
        ceil(ceil(N / W) / samples_per_gpu) collectives;
      * masks and panoptic maps never travel: with `annotations` + `evaluator`
        (`TripletEvaluator`) every rank matches ITS images against their ground truth on its
        own GPU, and only the per-image match lists (a few KB of Python lists) are gathered
        once at the end; rank 0 adds them to `metrics` (`SceneGraphMetrics`) in dataset order.

    `dataset[i]` -> `(img, img_metas)` as `simple_test` takes them (one image), or
    `(decoded uint8 (H, W, 3) BGR image, None)`: the detector then runs the reference's test
    pipeline on the GPU in front of the backbone (`preprocess.TestPipeline`);
    `annotations[i]` -> dict(gt_rels, gt_labels, gt_masks) (or None).  `calibrate`: on the
    first call per detector, choose the stream -> hardware-queue placement of its pipeline on
    this rank's first batch (`PSGTr.calibrate_pipeline`: ~50 throw-away submissions, worth ~8 %
    of the step).  Returns a dict:
    `records` [N, L] float32 in dataset order on EVERY rank (`unpack_triplets` splits a row),
    `num_images`, `world_size`, `rank`, `collectives`, and on rank 0 `metrics`
    (`metrics.summary()`) when an evaluator was given."""
    ini = dist.is_available() and dist.is_initialized()
    W = dist.get_world_size(group) if ini else 1
    rank = dist.get_rank(group) if ini else 0
    N = len(dataset)
    k = int(samples_per_gpu)
    if k < 1:
        raise ValueError("samples_per_gpu >= 1")
    steps = ((N + W - 1) // W + k - 1) // k
    mine = shard_indices(N, rank, W)
    groups = [mine[j:j + k] for j in range(0, len(mine), k)]
    col = TripletCollector(detector.bbox_head, depth=depth, n_local=k, group=group,
                           host_staging=host_staging, force_collective=force_collective,
                           keep_steps=steps)
    if calibrate and mine and hasattr(detector, "calibrate_pipeline") and \
            hasattr(detector, "_pipelines") and detector._pipelines() and \
            not detector.pipeline_calibrated(depth):
        detector.calibrate_pipeline(*collate([dataset[i] for i in groups[0]]), depth=depth)
    local_evals = []
    batches = (collate([dataset[i] for i in grp]) for grp in groups)
    done = 0
    for tb in detector.stream_triplets(batches, rescale=rescale, depth=depth):
        grp = groups[done]
        if len(tb.results) != len(grp):
            raise ValueError("the detector returned %d results for a batch of %d images"
                             % (len(tb.results), len(grp)))
        anns = [annotations[idx] if annotations is not None else None for idx in grp]
        # a lazily built annotation (`dataset.eval_ground_truth`: H2D copy + k_pan_masks) is
        # produced on the CALLER's stream, the evaluator reads it on the chain stream that
        # produced the result: order the two
        ann_ready = None
        if evaluator is not None and col.on_gpu and any(a is not None for a in anns):
            ann_ready = torch.cuda.Event()
            ann_ready.record(torch.cuda.current_stream(col.device))

        def evaluate(tb, grp=grp, anns=anns, ann_ready=ann_ready):
            if evaluator is None:
                return
            if ann_ready is not None:
                torch.cuda.current_stream(col.device).wait_event(ann_ready)
            for res, idx, ann in zip(tb.results, grp, anns):
                if ann is None:
                    continue
                ev = evaluator(res, ann["gt_rels"], ann["gt_labels"], ann["gt_masks"])
                iou = evaluator.iou_stats(res, ann["gt_rels"], ann["gt_labels"],
                                          ann["gt_masks"]) \
                    if hasattr(evaluator, "iou_stats") and len(ann["gt_rels"]) else None
                local_evals.append((idx, ev, ann["gt_rels"], iou))
        col.add(tb, before_release=evaluate)
        done += 1
    for _ in range(steps - done):          # ranks with fewer batches: zero rows
        col.pad()
    records = col.finish()
    if records is None:                    # an empty dataset: no steps, no records
        records = torch.zeros((0, col.gatherer.L), dtype=torch.float32)
    out = dict(records=records[:N], num_images=N, world_size=W, rank=rank,
               collectives=col.gatherer.gathered, local_indices=mine)
    if evaluator is not None:
        evals = [local_evals]
        if W > 1:
            evals = [None] * W
            dist.all_gather_object(evals, local_evals, group=group)
        if rank == 0 and metrics is not None:
            for idx, ev, gt_rels, iou in sorted((e for part in evals for e in part),
                                                key=lambda e: e[0]):
                metrics.add(ev, gt_rels, iou=iou)
            out["metrics"] = metrics.summary()
    return out


class GradReducer:
    """Data-parallel gradient reduction for the training step (what mmcv's MMDistributedDataParallel
    does behind tools/train.py:115-241): the gradients live in ONE flat buffer laid out in the
    order the backward pass completes them (`RelationTailGrad.flat_grad`), cut into buckets of
    `bucket_bytes`; `ready(end)` -- the backward pass's hook -- starts the all-reduce (SUM; RCCL
    over xGMI for backend "nccl") of every bucket that lies wholly inside flat[:end] on a side
    stream, behind an event on the computing stream, so the collectives of the early buckets run
    under the rest of the backward pass; `finish()` orders the computing stream behind all of them.
    The average is not a pass over the buffer: `scale` (= 1 / world) is what the consumer
    multiplies by where it reads the gradient (`pn_grad_norm_clip_f32` / `pn_adamw_f32`'s `pre`).

    Bucket size: xGMI is point to point, a ring all-reduce is bound per link, and a collective's
    fixed cost is tens of microseconds -- so few, large buckets (default 32 MiB: the 41 MB of the
    Pair-Net tail are two collectives), not the 25 MB / many-bucket habit of NVSwitch nodes.
    With one rank (or no process group) every call is a no-op and `scale` is 1."""

    def __init__(self, flat, group=None, bucket_bytes=32 << 20, force_collective=False):
        ini = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if ini else 1
        self.active = ini and (self.world > 1 or force_collective)
        self.flat, self.group = flat, group
        self.scale = 1.0 / self.world
        per = max(64, int(bucket_bytes) // 4 // 64 * 64)
        n = flat.numel()
        self.bounds = [(b, min(b + per, n)) for b in range(0, n, per)]
        self.on_gpu = flat.is_cuda
        self.side = torch.cuda.Stream(flat.device) if (self.on_gpu and self.active) else None
        self.next, self.works = 0, []
        self.collectives = 0

    def start(self):
        """Begin a new backward pass (the buffer is about to be zeroed and refilled)."""
        if self.works:
            raise RuntimeError("GradReducer.start(): the previous pass was not finish()ed")
        self.next = 0      # (finish() ordered the computing stream behind the last pass's reads)

    def ready(self, end):
        if not self.active:
            return
        ev = None
        while self.next < len(self.bounds) and self.bounds[self.next][1] <= end:
            b, e = self.bounds[self.next]
            view = self.flat[b:e]
            if self.side is not None:
                if ev is None:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.flat.device))
                    self.side.wait_event(ev)
                with torch.cuda.stream(self.side):
                    self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group,
                                                      async_op=True))
            else:
                self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group,
                                                  async_op=True))
            self.collectives += 1
            self.next += 1

    def finish(self):
        """Every bucket reduced; the caller's current stream is ordered behind the collectives."""
        if not self.active:
            return
        self.ready(self.flat.numel())
        for w in self.works:
            w.wait()
        if self.side is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.side)
        self.works = []
