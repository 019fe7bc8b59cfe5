"""Probe: what the simple_test leg (results to pinned host memory) costs, by variant."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import CrossHead2, PipelinedHead, ResNet50Hip, ResultStreamer, pairnet_head_cfg
dev = torch.device("cuda:0")
cfg = pairnet_head_cfg(); cfg.pop("type")
head = CrossHead2(**cfg); head.init_weights(seed=0); head.to(dev); head.use_graphs = True
net = ResNet50Hip().to(dev); net.use_graphs = True
H, W = 800, 1333
metas = [dict(img_shape=(H, W, 3), scale_factor=[2.083] * 4)]
g = torch.Generator().manual_seed(1)
pool = [torch.randn(1, 3, H, W, generator=g).to(dev) for _ in range(8)]


def run(variant, ring, depth=4, n=60):
    pipe = PipelinedHead(head, depth=depth, a_streams=2)
    net.grid_reserve = pipe.grid_reserve
    st = None if variant == "none" else ResultStreamer(head, ring=ring, stage_on_device=variant == "staged")
    cnt = [0]

    def take(res):
        if st is None:
            return
        if len(st) >= st.ring - 1:
            st.pop()
        st.push(res, pipe)

    def steps(k):
        for _ in range(k):
            sl = pipe.count % 2
            with torch.cuda.stream(pipe.streams_a[sl]):
                r = pipe.submit(net(pool[cnt[0] % 8], slot=sl), metas)
                cnt[0] += 1
                if r is not None:
                    take(r)
        with torch.cuda.stream(pipe.streams_a[0]):
            for r in pipe.flush():
                take(r)
        while st is not None and len(st):
            st.pop()
    steps(3 * depth)
    torch.cuda.synchronize()
    t = time.perf_counter()
    steps(n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("%-8s ring %d depth %d: %.3f ms/step  %.1f images/s" % (variant, ring, depth, 1e3 * dt, 1 / dt), flush=True)


for variant, ring, depth in (("none", 0, 4), ("direct", 6, 4), ("staged", 6, 4), ("staged", 3, 4),
                             ("direct", 8, 6), ("staged", 8, 5), ("none", 0, 4)):
    run(variant, ring, depth)
