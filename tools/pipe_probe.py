"""Nothing but the headline loop (for rocprofv3 timelines): N pipelined image -> triplets steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import CrossHead2, PipelinedHead, ResNet50Hip, pairnet_head_cfg
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(os.environ.get("BATCH", 1))
cfg = pairnet_head_cfg(); cfg.pop("type")
head = CrossHead2(**cfg); head.init_weights(seed=0); head.to(dev); head.use_graphs = True
if os.environ.get("EXACT"):     # EXACT=0 / 1 / full: attention-mask operation order
    head.exact_mask_order = {"0": False, "1": True}.get(os.environ["EXACT"], os.environ["EXACT"])
net = ResNet50Hip().to(dev); net.use_graphs = True
H, W = 800, 1333
metas = [dict(img_shape=(H, W, 3), scale_factor=[2.083] * 4)] * B
g = torch.Generator().manual_seed(1)
pool = [torch.randn(B, 3, H, W, generator=g).to(dev) for _ in range(4)]
pipe = PipelinedHead(head, depth=int(os.environ.get("DEPTH", 4)), a_streams=int(os.environ.get("ASTREAMS", 2)),
                     grid_trim=int(os.environ.get("RESERVE", 64)))
net.grid_reserve = pipe.grid_reserve
cnt = [0]


def one():
    sl = pipe.count % len(pipe.streams_a)
    with torch.cuda.stream(pipe.streams_a[sl]):
        pipe.submit(net(pool[cnt[0] % 4], slot=sl), metas)
    cnt[0] += 1


for _ in range(12):
    one()
pipe.flush()
if not os.environ.get("NOCAL"):
    pipe.calibrate(None, metas, submit=one)
for _ in range(8):
    one()
pipe.flush()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(N):
    one()
pipe.flush()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / N
print("%.3f ms/step, %.1f images/s" % (1e3 * dt, B / dt))
