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
        self.filled = 0                       # steps packed so far
        self.gathered = 0                     # steps gathered so far

    # ---- ring form -----------------------------------------------------------------------
    def begin_step(self):
        """Select the ring entry of a new step: pack() then writes there (on the caller's
        current stream); finish with end_step()."""
        if self.filled - self.gathered >= self.ring:
            raise RuntimeError("TripletGatherer ring is full: gather_delayed() / flush() first")
        self.send = self.sends[self.filled % self.ring]

    def end_step(self):
        if self.on_device:
            ev = self.packed[self.filled % self.ring] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.packed[self.filled % self.ring] = ev
        self.filled += 1

    def gather_delayed(self, delay, host_staging=False):
        """All-gather the oldest packed step if it is at least `delay` steps old (on the
        caller's current stream, behind that step's `packed` event); returns its records or
        None."""
        if self.filled - self.gathered <= delay:
            return None
        k = self.gathered % self.ring
        if self.on_device and self.packed[k] is not None:
            torch.cuda.current_stream().wait_event(self.packed[k])
        self.send = self.sends[k]
        out = self.gather(host_staging=host_staging)
        self.gathered += 1
        return out

    def flush(self, host_staging=False):
        """Gather every step still in the ring (end of the run)."""
        outs = []
        while self.gathered < self.filled:      # (copies: gather() returns an internal buffer)
            outs.append(self.gather_delayed(0, host_staging=host_staging).clone())
        return outs

    def pack(self, i, labels, rel_dists, sub_pos, obj_pos):
        self.hip.pack_triplets(labels, rel_dists, sub_pos, obj_pos, self.send[i], self.R, self.C1)

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
