"""How much the query chains cost the chip-filling stream: the headline loop with stage B cut down."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import CrossHead2, PipelinedHead, ResNet50Hip, pairnet_head_cfg
dev = torch.device("cuda:0")
cfg = pairnet_head_cfg(); cfg.pop("type")
H, W = 800, 1333
metas = [dict(img_shape=(H, W, 3), scale_factor=[2.083] * 4)]
g = torch.Generator().manual_seed(1)
pool = [torch.randn(1, 3, H, W, generator=g).to(dev) for _ in range(4)]


def run(name, patch):
    head = CrossHead2(**cfg); head.init_weights(seed=0); head.to(dev); head.use_graphs = True
    net = ResNet50Hip().to(dev); net.use_graphs = True
    patch(head)
    pipe = PipelinedHead(head, depth=4, a_streams=2)
    net.grid_reserve = pipe.grid_reserve
    cnt = [0]

    def one():
        sl = pipe.count % 2
        with torch.cuda.stream(pipe.streams_a[sl]):
            pipe.submit(net(pool[cnt[0] % 4], slot=sl), metas)
        cnt[0] += 1
    for _ in range(12):
        one()
    pipe.flush()
    pipe.calibrate(None, metas, submit=one)
    for _ in range(8):
        one()
    pipe.flush()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(100):
        one()
    pipe.flush()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 100
    print("%-44s %.3f ms/step  %.1f images/s" % (name, 1e3 * dt, 1 / dt), flush=True)


def full(h):
    pass


def no_b(h):
    h._stage_b = lambda pl: None


def no_rel(h):
    h._relation_stage = lambda pl: None


def obj_only_3(h):
    orig = h._object_decoder
    h.num_dec_layers_saved = h.num_dec_layers
    def f(pl, *a, **k):
        n = h.num_dec_layers
        h.num_dec_layers = 3
        try:
            return orig(pl, *a, **k)
        finally:
            h.num_dec_layers = n
    h._object_decoder = f
    h._relation_stage = lambda pl: None


def no_post(h):
    h.get_bboxes = lambda *a, **k: CrossHead2.ResultList([])


run("full", full)
run("no stage B (chains), get_bboxes kept", no_b)
run("object decoder only (no PPN / relation)", no_rel)
run("3 decoder layers only", obj_only_3)
run("full", full)
