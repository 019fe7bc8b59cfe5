"""Two-stream software pipeline over consecutive batches.

The head has two very different halves (head.py): stage A (pixel decoder + K/V
projections) is a few dozen chip-filling GEMM / gather launches, stage B (the
sequential query chain of the two decoders, PPN, top-k, post-processing) is ~200
latency-bound launches that each occupy a handful of CUs.  Run back to back, stage B
leaves most of the 256 CUs idle.  `PipelinedHead` runs stage B of batch i on one HIP
stream while stage A of batch i+1 runs on another, with two buffer sets (`slot` 0/1)
so the stages never share mutable state:

    stream A:  A(0) A(1)       A(2)       A(3) ...
    stream B:       B(0)+post  B(1)+post  B(2)+post ...

Results are those of `CrossHead2.simple_test_bboxes`, returned one submission late;
`flush()` returns the last one.  Images are independent (pairnet_head.py:260-417 has
no cross-image op), so this is the same computation, only scheduled for throughput.
"""
import torch


class PipelinedHead:
    def __init__(self, head):
        if head.device is None or head.device.type != "cuda":
            raise RuntimeError("PipelinedHead needs a head on an MI355X (.to('cuda:N'))")
        self.head = head
        with torch.cuda.device(head.device):
            # stage B's small dependent kernels must not queue behind stage A's thousands
            # of workgroups: give its stream the high hardware-queue priority
            self.stream_a = torch.cuda.Stream(priority=0)
            self.stream_b = torch.cuda.Stream(priority=-1)
            self.a_done = [torch.cuda.Event(), torch.cuda.Event()]
            self.b_done = [torch.cuda.Event(), torch.cuda.Event()]
        self.count = 0
        self.pending = None       # (slot, plan, img_metas, rescale) whose stage B has not run

    @torch.no_grad()
    def submit(self, feats, img_metas, rescale=False):
        """Queue one batch; returns the previous batch's result list (None the first time)."""
        head = self.head
        B, shapes, hw2 = head._check_feats(feats, img_metas)
        slot = self.count & 1
        pl = head._plan(B, shapes, hw2, slot)
        cur = torch.cuda.current_stream(head.device)
        self.stream_a.wait_stream(cur)               # feats produced on the caller's stream
        if self.count >= 2:
            self.stream_a.wait_event(self.b_done[slot])   # slot's buffers free again
        with torch.cuda.stream(self.stream_a):
            head._run_stage("a", pl, feats)
            self.a_done[slot].record(self.stream_a)
        for f in feats:                               # keep feats alive until A has read them
            f.record_stream(self.stream_a)
        prev = self._finish()
        self.pending = (slot, pl, img_metas, rescale)
        self.count += 1
        return prev

    def _finish(self):
        if self.pending is None:
            return None
        head = self.head
        slot, pl, metas, rescale = self.pending
        self.pending = None
        with torch.cuda.stream(self.stream_b):
            self.stream_b.wait_event(self.a_done[slot])
            head._run_stage("b", pl)
            head._last_plan = pl
            res = head.get_bboxes(*head._outputs(pl), metas, rescale=rescale)
            self.b_done[slot].record(self.stream_b)
        # the caller consumes the results on its own stream
        cur = torch.cuda.current_stream(head.device)
        cur.wait_event(self.b_done[slot])
        for tup in res:
            for t in tup:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(cur)
        return res

    @torch.no_grad()
    def flush(self):
        """Finish the batch still in flight and return its results (or None)."""
        res = self._finish()
        torch.cuda.current_stream(self.head.device).wait_stream(self.stream_b)
        return res
