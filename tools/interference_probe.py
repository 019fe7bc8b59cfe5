"""Tuning aid: what in the query chains (stage B) costs the pixel decoder (stage A) time when
they overlap in the 3-deep pipeline.  Stage B is replaced by cut-down variants (timing only:
the outputs of the cut variants are meaningless) and the pipelined step time is measured."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import CrossHead2, PipelinedHead, pairnet_head_cfg
dev = torch.device("cuda:0")
shapes = [(200, 334), (100, 167), (50, 84), (25, 42)]
feats = [torch.relu(torch.randn(1, c, h, w)).to(dev) for c, (h, w) in zip((256, 512, 1024, 2048), shapes)]
metas = [dict(img_shape=(800, 1333, 3), scale_factor=[2.083] * 4)]

def make(variant):
    class H(CrossHead2):
        def _stage_b(self, pl):
            if variant == "full":
                return CrossHead2._stage_b(self, pl)
            if variant == "none":
                return
            if variant == "relation_only":
                return self._relation_stage(pl)
            if variant == "object_only":
                return self._object_decoder(pl)
            if variant.startswith("layers"):
                n, self.num_dec_layers = self.num_dec_layers, int(variant[6:])
                try:
                    self._object_decoder(pl)
                finally:
                    self.num_dec_layers = n
        def get_bboxes(self, *a, **k):
            return [()] if variant != "full" else CrossHead2.get_bboxes(self, *a, **k)
    cfg = pairnet_head_cfg(); cfg.pop("type")
    h = H(**cfg); h.init_weights(seed=0); h.to(dev); h.use_graphs = True
    return h

for variant in ("full", "none", "relation_only", "object_only", "layers3", "layers6"):
    head = make(variant)
    eng = PipelinedHead(head, depth=3)
    def steps(n):
        for _ in range(n): eng.submit(feats, metas)
        eng.flush()
    steps(8)
    eng.calibrate(feats, metas)
    best = None
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        steps(40)
        torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t) / 40
        best = dt if best is None else min(best, dt)
    print("stage B = %-14s  %.3f ms/step" % (variant, best))
    del eng, head
