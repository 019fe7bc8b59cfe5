"""Multi-stream software pipeline over consecutive batches.

The head has two very different halves (head.py): stage A (pixel decoder + K/V
projections) is a few dozen chip-filling GEMM / gather launches; stage B (the
sequential query chain of the two decoders, PPN, top-k) is ~200 latency-bound launches
that each occupy a handful of CUs, followed by the post-processing (`get_bboxes`).
Run back to back, stage B leaves most of the 256 CUs idle; run beside stage A, each of
its small kernels queues behind resident GEMM workgroups, so its chain stretches to
about twice its stand-alone time.  `PipelinedHead(depth=4, a_streams=2)` therefore keeps
FOUR batches in flight on four HIP streams and four buffer sets ("slots"):

    stream A0:  [backbone(i)]   A(i)      producer + stage A of the even batches ...
    stream A1:  [backbone(i-1)] A(i-1)    ... and of the odd ones: two chip-filling kernel
                                          sequences side by side, so that the HBM-bound
                                          kernels of one (norms, resampling, deformable
                                          sampling, Winograd transforms) and its launch
                                          ramps / tails overlap the other's MFMA GEMMs
    stream B0:  B(i-2)                    query chain of an older batch ...
    stream B1:  B(i-3), get_bboxes(i-3)   ... and of the oldest, whose results submit(i)
                                          returns (its host syncs only wait for work
                                          queued three submissions ago)

(Measured on MI355X, image tensor -> triplets: 160 images/s with one stage-A stream and
depth 3, 189 with two and depth 4; two stage-A streams with ONE chain stream -- depth 3 --
lose: 145.)

Results are exactly those of `CrossHead2.simple_test_bboxes` (bitwise: scheduling
only), returned `depth-1` submissions late; `flush()` drains the rest.  Images are
independent (pairnet_head.py:260-417 has no cross-image op).

Result lifetime: the returned tensors are views of the slot's buffers, ordered behind the
stream `submit()` / `flush()` was called on; work queued on THAT stream before the next
`submit()` is safe.  A caller that reads them on another stream (a D2H copy stream, the
all-gather's side stream) says so with `consumed(results, stream)`: the slot is not
reused before that stream has passed the recorded point.
"""
import torch

from .hip import on_device


class PipelinedHead:
    def __init__(self, head, depth=4, a_streams=2, grid_trim=64):
        if head.device is None or head.device.type != "cuda":
            raise RuntimeError("PipelinedHead needs a head on an MI355X (.to('cuda:N'))")
        if depth < 2 or not 1 <= a_streams < depth:
            raise ValueError("depth >= 2 and 1 <= a_streams < depth")
        self.head, self.depth, self.device = head, depth, head.device
        with torch.cuda.device(head.device):
            # a_streams > 1: stage A of consecutive batches alternates between streams, so
            # one batch's gather / normalisation kernels can run beside another's GEMMs
            self.streams_a = [torch.cuda.Stream(priority=0) for _ in range(a_streams)]
            # same priority for the query chains (measured: 221.6 images/s against 218.1
            # with the chains on the high-priority queues)
            self.streams_b = [torch.cuda.Stream(priority=0)
                              for _ in range(max(1, depth - a_streams))]
            self.a_done = [torch.cuda.Event() for _ in range(depth)]
            self.b_done = [torch.cuda.Event() for _ in range(depth)]
        self.read_done = [[] for _ in range(depth)]   # per slot: events of foreign-stream readers
        self.count = 0
        self.queue = []   # per in-flight batch: dict(slot, pl, metas, rescale, b_started)
        # The persistent GEMM kernels of stage A fill every workgroup slot of the chip, so a
        # query-chain kernel of the other streams only gets on at a kernel boundary.  Leaving
        # 64 of the 1024 slots free lets the chains run beside stage A (measured with one
        # stage-A stream: 158.2 / 160.5 / 160.4 images/s at 0 / 32 / 48 free slots, 154.7 at
        # 256; with two: 188.2 / 189.5 / 189.7 at 0 / 64 / 128).  A property of THIS head's
        # stage-A launches (a per-call hint in the GEMM descriptors, hip.reserve_slots) --
        # nothing process-wide: other heads / pipelines in the process keep their own grids.
        # A producer the caller runs in front of stage A (the backbone) takes the same hint
        # through its own `grid_reserve` attribute.
        self.grid_reserve = grid_trim
        head.grid_reserve = grid_trim

    @torch.no_grad()
    def calibrate(self, feats, img_metas, steps=8, submit=None):
        """Pick the stream -> hardware-queue placement empirically.  HIP spreads streams
        round-robin over (by default) 4 hardware queues, and which of them carries stage A
        and which the query chains changes the pipelined step time by ~8 % on MI355X
        (4.5 vs 4.9 ms measured; the rotations alternate fast / slow).  Four consecutively
        created streams sit on the four queues; every rotation of the roles over them is
        timed on `steps` batches of the given input and the fastest one is kept.  Call it
        once during warm-up (the pipeline must be empty); returns the per-rotation times in
        ms per batch.  `submit`: a callable that queues ONE batch the way the caller's loop
        does (e.g. with its backbone in front, on the stage-A stream of that batch), so that
        the placement is chosen for the real workload; default: `self.submit(feats, metas)`."""
        import time
        if self.queue:
            raise RuntimeError("calibrate() needs an empty pipeline: call flush() first")
        dev = self.head.device
        with torch.cuda.device(dev):
            pool = [torch.cuda.Stream(priority=0) for _ in range(4)]
        na, nb = len(self.streams_a), len(self.streams_b)
        times, best = [], None
        import itertools, os
        cands = [[(r + i) % 4 for i in range(na + nb)] for r in range(4)]
        if os.environ.get("PAIRNET_CALIBRATE_ALL"):
            cands = [list(p) for p in itertools.permutations(range(4), min(4, na + nb))]
        self.calibration_candidates = cands
        for r, cand in enumerate(cands):
            order = [pool[i] for i in cand]
            self.streams_a, self.streams_b = order[:na], order[na:]
            one = submit if submit is not None else (lambda: self.submit(feats, img_metas))
            for _ in range(3):
                one()
            self.flush()
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            for _ in range(steps):
                one()
            self.flush()
            torch.cuda.synchronize(dev)
            times.append(1e3 * (time.perf_counter() - t) / steps)
            if best is None or times[-1] < times[best]:
                best = r
        order = [pool[i] for i in cands[best]]
        self.streams_a, self.streams_b = order[:na], order[na:]
        self.calibration_ms = times
        return times

    def _stream_b(self, index):
        return self.streams_b[index % len(self.streams_b)]

    @torch.no_grad()
    @on_device
    def submit(self, feats, img_metas, rescale=False):
        """Queue one batch; returns the result list of the batch submitted depth-1 calls
        earlier (None while the pipeline fills).  The returned tensors are ordered behind
        the stream submit() was called on (torch's current stream): consume them there."""
        head = self.head
        B, shapes, hw2 = head._check_feats(feats, img_metas)
        idx = self.count
        slot = idx % self.depth
        pl = head._plan(B, shapes, hw2, slot, head._feats_nhwc)
        cur = torch.cuda.current_stream(head.device)
        sa = self.streams_a[idx % len(self.streams_a)]
        sa.wait_stream(cur)                                # feats produced on the caller's stream
        if idx >= self.depth:
            sa.wait_event(self.b_done[slot])               # slot's buffers are free again
        for ev in self.read_done[slot]:                    # ... also for foreign-stream readers
            sa.wait_event(ev)
        self.read_done[slot] = []
        with torch.cuda.stream(sa):
            head._run_stage("a", pl, feats)
            self.a_done[slot].record(sa)
        # a producer that reuses its output buffers (the native backbone does) must not
        # overwrite them before stage A has read them
        cur.wait_event(pl.feats_read)
        for f in feats:                                    # keep feats alive until A has read them
            f.record_stream(sa)
        self.queue.append(dict(idx=idx, slot=slot, pl=pl, metas=img_metas, rescale=rescale,
                               b_started=False))
        self.count += 1
        # start the query chain of every batch whose stage A is not among the newest ones
        for item in self.queue[:len(self.queue) - len(self.streams_a)]:
            self._start_b(item)
        if len(self.queue) >= self.depth:
            return self._finish(self.queue.pop(0))
        return None

    def _start_b(self, item):
        if item["b_started"]:
            return
        sb = self._stream_b(item["idx"])
        with torch.cuda.stream(sb):
            sb.wait_event(self.a_done[item["slot"]])
            self.head._run_stage("b", item["pl"])
        item["b_started"] = True

    def _finish(self, item):
        head = self.head
        self._start_b(item)
        sb = self._stream_b(item["idx"])
        with torch.cuda.stream(sb):
            head._last_plan = item["pl"]
            res = head.get_bboxes(*head._outputs(item["pl"]), item["metas"],
                                  rescale=item["rescale"])
            self.b_done[item["slot"]].record(sb)
        # the caller consumes the results on its own stream
        cur = torch.cuda.current_stream(head.device)
        cur.wait_event(self.b_done[item["slot"]])
        for tup in res:
            for t in tup:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(cur)
        try:
            res.pipeline_slot = item["slot"]
        except AttributeError:                 # a plain list: give it a place for the tag
            from .head import CrossHead2
            res = CrossHead2.ResultList(res)
            res.pipeline_slot = item["slot"]
        # the chain stream the results were produced on: a consumer that queues its reads
        # THERE (ResultStreamer) is ordered behind get_bboxes without waiting for whatever
        # the caller has queued on its own stream since
        res.pipeline_stream = sb
        return res

    def consumed(self, results, stream=None):
        """Declare that `results` (a list submit() / flush() returned) are read on `stream`
        (default: the current one) up to this point: the slot's buffers are not overwritten
        before that stream gets here.  Needed only for readers on a stream other than the one
        the results were returned on."""
        slot = getattr(results, "pipeline_slot", None)
        if slot is None:
            raise ValueError("consumed(): not a result list returned by submit() / flush()")
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        ev = torch.cuda.Event()
        ev.record(stream)
        self.read_done[slot].append(ev)

    @torch.no_grad()
    @on_device
    def flush(self):
        """Finish every batch still in flight; returns the list of their result lists
        (oldest first)."""
        out = []
        while self.queue:
            out.append(self._finish(self.queue.pop(0)))
        return out
