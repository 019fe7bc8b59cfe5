"""Tuning aid: how much of the pipelined step is owed to stage B?  Times the pipelined
loop with the relation decoder's layers removed (wrong results, timing only)."""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import CrossHead2, PipelinedHead, pairnet_head_cfg
dev = torch.device("cuda:0")
shapes = [(200, 334), (100, 167), (50, 84), (25, 42)]
feats = [torch.relu(torch.randn(1, c, h, w)).to(dev) for c, (h, w) in zip((256, 512, 1024, 2048), shapes)]
metas = [dict(img_shape=(800, 1333, 3), scale_factor=[2.083] * 4)]
def run(rel_layers, skip_b=False):
    cfg = pairnet_head_cfg(); cfg.pop("type")
    head = CrossHead2(**cfg); head.init_weights(seed=0); head.to(dev); head.use_graphs = True
    head.num_rel_layers = rel_layers
    if skip_b:
        head._stage_b = lambda pl: None
    eng = PipelinedHead(head, depth=3)
    for _ in range(10): eng.submit(feats, metas)
    eng.flush(); torch.cuda.synchronize()
    gc.collect()
    t = time.perf_counter()
    for _ in range(40): eng.submit(feats, metas)
    eng.flush(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / 40
print("no stage B at all (stage A + get_bboxes): %.3f ms/step" % run(6, True))

# which stage-B kernels hurt?  (wrong results, timing only)
from pairnet_amd import hip
def run2(skip):
    cfg = pairnet_head_cfg(); cfg.pop("type")
    head = CrossHead2(**cfg); head.init_weights(seed=0); head.to(dev); head.use_graphs = True
    orig_gemm, orig_gather = hip.gemm, hip.gather_rows
    def gemm(A, W, C, **kw):
        if "mp" in skip and kw.get("N", 0) >= 60000: return
        if "mask2" in skip and kw.get("N", 0) == 16700 and kw.get("M", 0) == 100: return
        return orig_gemm(A, W, C, **kw)
    def gather(x, index, out, B, rows_in, rows_out, length):
        if "gather" in skip and length >= 60000: return
        return orig_gather(x, index, out, B, rows_in, rows_out, length)
    hip.gemm, hip.gather_rows = gemm, gather
    if "post" in skip:
        head.get_bboxes = lambda *a, **k: [()]
    eng = PipelinedHead(head, depth=3)
    for _ in range(10): eng.submit(feats, metas)
    eng.flush(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(40): eng.submit(feats, metas)
    eng.flush(); torch.cuda.synchronize()
    hip.gemm, hip.gather_rows = orig_gemm, orig_gather
    return 1e3 * (time.perf_counter() - t) / 40
for skip in (set(),):
    print("skip %-32s %s ms/step" % (sorted(skip), " ".join("%.3f" % run2(skip) for _ in range(8))))
