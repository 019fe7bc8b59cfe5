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
pipe = PipelinedHead(head, depth=4, a_streams=2)
net.grid_reserve = pipe.grid_reserve
cnt = [0]
take = [None]


def one():
    sl = pipe.count % 2
    with torch.cuda.stream(pipe.streams_a[sl]):
        r = pipe.submit(net(pool[cnt[0] % 8], slot=sl), metas)
        cnt[0] += 1
        if r is not None and take[0] is not None:
            take[0](r)


def flush():
    with torch.cuda.stream(pipe.streams_a[0]):
        for r in pipe.flush():
            if take[0] is not None:
                take[0](r)


for _ in range(12):
    one()
flush()
pipe.calibrate(None, metas, submit=one)


def run(name, mk, n=100):
    st = mk()
    host = [0.0]

    def tk(res):
        t = time.perf_counter()
        if st is not None:
            if len(st) >= st.ring - 1:
                st.pop()
            st.push(res, pipe)
        host[0] += time.perf_counter() - t
    take[0] = tk if st is not None else None
    for _ in range(10):
        one()
    flush()
    while st is not None and len(st):
        st.pop()
    host[0] = 0.0
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        one()
    flush()
    while st is not None and len(st):
        st.pop()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("%-28s %.3f ms/step  %.1f images/s  (host in push/pop %.3f ms/step)" % (name, 1e3 * dt, 1 / dt, 1e3 * host[0] / n), flush=True)


class ThrottleOnly(ResultStreamer):   # events + host pacing, no copies at all
    def push(self, results, pipe=None):
        key = ("x",)
        e = self.entries[self.head_i % self.ring]
        if e is None:
            e = self.entries[self.head_i % self.ring] = dict(key=key, event=torch.cuda.Event())
        cur = torch.cuda.current_stream()
        if MODE == "copystream":
            self.stream.wait_stream(cur)
            e["event"].record(self.stream)
        else:
            e["event"].record(cur)
        self.head_i += 1

    def pop(self):
        e = self.entries[self.tail_i % self.ring]
        self.tail_i += 1
        e["event"].synchronize()
        return []


# one variant per process (allocations of earlier variants shift the later ones' numbers):
#   d2h_probe.py none | pacing | <copy_wgs> (masks as bits) | <copy_wgs>u (masks as bytes)
which = sys.argv[1] if len(sys.argv) > 1 else "4"
MODE = "cur"
if which == "none":
    run("no result copies", lambda: None)
elif which == "pacing":
    run("host pacing only (event on cur)", lambda: ThrottleOnly(head, ring=6))
else:
    pack = not which.endswith("u")
    w = int(which.rstrip("u"))
    run(("staged, per field, copy kernel %d WGs" % w if w else "staged, per field, hipMemcpyAsync") +
        (", masks as bits" if pack else ", masks as bytes"),
        lambda: ResultStreamer(head, ring=6, copy_wgs=w, pack_masks=pack))
