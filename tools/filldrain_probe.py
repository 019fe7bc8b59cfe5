"""Fill / drain cost of the pipelined loop: wall time of N synchronised-at-both-ends steps for
several N (T = overhead + N * interval), and the completion time of every image of one run."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import CrossHead2, PipelinedHead, ResNet50Hip, pairnet_head_cfg
dev = torch.device("cuda:0")
cfg = pairnet_head_cfg(); cfg.pop("type")
head = CrossHead2(**cfg); head.init_weights(seed=0); head.to(dev); head.use_graphs = True
net = ResNet50Hip().to(dev); net.use_graphs = True
H, W = 800, 1333
metas = [dict(img_shape=(H, W, 3), scale_factor=[2.083] * 4)]
g = torch.Generator().manual_seed(1)
pool = [torch.randn(1, 3, H, W, generator=g).to(dev) for _ in range(4)]
pipe = PipelinedHead(head, depth=int(os.environ.get("DEPTH", 4)), a_streams=int(os.environ.get("ASTREAMS", 2)))
net.grid_reserve = pipe.grid_reserve
cnt = [0]
done = []


def one(mark=False):
    sl = pipe.count % len(pipe.streams_a)
    with torch.cuda.stream(pipe.streams_a[sl]):
        res = pipe.submit(net(pool[cnt[0] % 4], slot=sl), metas)
    cnt[0] += 1
    return res


for _ in range(12):
    one()
pipe.flush()
pipe.calibrate(None, metas, submit=one)
for _ in range(8):
    one()
pipe.flush()
import gc
for idle in (0.0, 0.05, 0.2, 1.0):
    torch.cuda.synchronize(); time.sleep(idle)
    t = time.perf_counter()
    for _ in range(20):
        one()
    pipe.flush(); torch.cuda.synchronize()
    print("after %.2f s idle: 20 steps in %.2f ms" % (idle, 1e3 * (time.perf_counter() - t)), flush=True)
for N in (20, 40):
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(N):
            one()
        t_sub = time.perf_counter() - t
        pipe.flush()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t, t_sub))
    best = min(ts)
    print("   reps: " + " ".join("%.2f" % (1e3 * a) for a, _ in ts))
    print("N=%3d: %.3f ms total, %.3f ms/step, host submit %.3f ms" % (N, 1e3 * best[0], 1e3 * best[0] / N, 1e3 * best[1]), flush=True)
