"""Tuning aid: is the pipelined loop CPU-bound?  Host enqueue time per step vs wall."""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import CrossHead2, PipelinedHead, pairnet_head_cfg
dev = torch.device("cuda:0")
cfg = pairnet_head_cfg(); cfg.pop("type")
head = CrossHead2(**cfg); head.init_weights(seed=0); head.to(dev); head.use_graphs = True
H, W = 800, 1333
shapes = [(200, 334), (100, 167), (50, 84), (25, 42)]
feats = [torch.relu(torch.randn(1, c, h, w)).to(dev) for c, (h, w) in zip((256, 512, 1024, 2048), shapes)]
metas = [dict(img_shape=(H, W, 3), scale_factor=[2.083] * 4)]
eng = PipelinedHead(head, depth=3)
for _ in range(10): eng.submit(feats, metas)
eng.flush(); torch.cuda.synchronize()
gc.collect(); gc.freeze(); gc.disable()
N = 40
t0 = time.perf_counter()
per = []
for _ in range(N):
    t = time.perf_counter(); eng.submit(feats, metas); per.append(time.perf_counter() - t)
t1 = time.perf_counter()
eng.flush(); torch.cuda.synchronize()
t2 = time.perf_counter()
per.sort()
print("host enqueue: %.3f ms/step (median %.3f, max %.3f); wall incl. drain: %.3f ms/step"
      % (1e3 * (t1 - t0) / N, 1e3 * per[N // 2], 1e3 * per[-1], 1e3 * (t2 - t0) / N))
# split: stage graphs vs get_bboxes
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): eng.submit(feats, metas)
pr.disable(); eng.flush(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
