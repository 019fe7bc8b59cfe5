"""Tuning aid (round 5): what the query chains (stage B) cost the IMAGE pipeline (backbone +
stage A on two streams, chains on two more) -- stage B replaced by cut-down variants (timing
only: the outputs of the cut variants are meaningless), pipelined ms per image at 800 x 1333."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import CrossHead2, PipelinedHead, ResNet50Hip, pairnet_head_cfg
dev = torch.device("cuda:0")
metas = [dict(img_shape=(800, 1333, 3), scale_factor=[2.083] * 4)]
g = torch.Generator().manual_seed(0)
pool = [torch.randn(1, 3, 800, 1333, generator=g).to(dev) for _ in range(4)]


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

        def get_bboxes(self, *a, **k):
            return [()] if variant != "full" else CrossHead2.get_bboxes(self, *a, **k)
    cfg = pairnet_head_cfg(); cfg.pop("type")
    h = H(**cfg); h.init_weights(seed=0); h.to(dev); h.use_graphs = True
    return h


for variant, trim in (("full", 64), ("none", 64), ("none", 0), ("object_only", 64),
                      ("relation_only", 64), ("full", 128), ("full", 32)):
    head = make(variant)
    net = ResNet50Hip().to(dev); net.use_graphs = True
    eng = PipelinedHead(head, depth=4, a_streams=2, grid_trim=trim)
    net.grid_reserve = eng.grid_reserve
    n = [0]

    def step():
        sl = eng.count % len(eng.streams_a)
        with torch.cuda.stream(eng.streams_a[sl]):
            eng.submit(net(pool[n[0] % len(pool)], slot=sl), metas)
        n[0] += 1

    def steps(k):
        for _ in range(k):
            step()
        with torch.cuda.stream(eng.streams_a[0]):
            eng.flush()
    steps(8)
    eng.calibrate(None, metas, submit=step)
    best = None
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        steps(60)
        torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t) / 60
        best = dt if best is None else min(best, dt)
    print("stage B = %-14s reserve %3d  %.3f ms/image" % (variant, trim, best), flush=True)
    del eng, head, net
    torch.cuda.empty_cache()
