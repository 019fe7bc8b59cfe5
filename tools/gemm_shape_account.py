"""Where the plain GEMM kernel's time goes: one eager image -> triplets step under the
per-launch HIP-event timer, launches grouped by (flops, bytes) = shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pairnet_amd import CrossHead2, ResNet50Hip, hip, pairnet_head_cfg

dev = torch.device("cuda:0")
cfg = pairnet_head_cfg(); cfg.pop("type")
head = CrossHead2(**cfg).to(dev)
bb = ResNet50Hip().to(dev)
img = torch.randn(1, 3, 800, 1333, device=dev)
metas = [dict(img_shape=(800, 1333, 3), scale_factor=[2.083] * 4)]
for _ in range(3):
    head.simple_test_bboxes(bb(img), metas)
REP = 5
hip.TIMER = hip.KernelTimer()
for _ in range(REP):
    head.simple_test_bboxes(bb(img), metas)
torch.cuda.synchronize()
agg = {}
for (name, flops, nbytes, s, e), meta in zip(hip.TIMER.records, hip.TIMER.meta):
    if "k_gemm_tile" not in name and "k_gemm_group" not in name:
        continue
    a = agg.setdefault((name + (" %dx%dx%d b%d%s" % (meta[:4] + ("s" if meta[4] else "",)) if meta else ""),
                        flops, nbytes), [0, 0.0])
    a[0] += 1; a[1] += s.elapsed_time(e)
hip.TIMER = None
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows) / REP
print("total %.3f ms/step over %d shapes" % (tot, len(rows)))
for (name, flops, nbytes), (n, ms) in rows:
    ms1 = ms / n
    print("%-46s x%-3d %7.1f us  %6.1f TF  %5.2f TB/s  %6.2f GF  %6.1f MB  (%.1f%%)" % (
        name[-44:], n // REP, ms1 * 1e3, flops / ms1 * 1e-9, nbytes / ms1 * 1e-9, flops * 1e-9,
        nbytes * 1e-6, 100 * ms / REP / tot))
