"""Debug: which images of a small shape mix differ between the pipelined product loop and an
eager single-stream call, per arithmetic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pairnet_amd import CrossHead2, pairnet_head_cfg
from pairnet_amd import ResNet50Hip
from pairnet_amd.detector import PSGTr
from pairnet_amd.dist import multi_gpu_test, pack_triplets
from pairnet_amd.preprocess import rescale_size

dev = "cuda:0"
arith = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
cfg = pairnet_head_cfg(); cfg.pop("type")
head = CrossHead2(**cfg); head.init_weights(seed=0); head.to(dev)
head.gemm_arithmetic = arith
backbone = ResNet50Hip(depth=50); backbone.to(dev)
det = PSGTr.from_parts(backbone, head)
ORIG = [(480, 640), (427, 640), (640, 480), (375, 500), (612, 612), (640, 640)]
gm = torch.Generator().manual_seed(77)
items = []
for k in range(36):
    h0, w0 = ORIG[k % len(ORIG)]
    hn, wn = rescale_size(h0, w0, (1333, 800))
    im = torch.randn(1, 3, hn, wn, generator=gm).to(dev)
    meta = dict(img_shape=(hn, wn, 3), ori_shape=(h0, w0, 3), pad_shape=(hn, wn, 3),
                scale_factor=[wn / w0, hn / h0, wn / w0, hn / h0])
    items.append((im, [meta]))
det.reserve([(800, 1333), (1333, 800)], depth=4, orig_sizes=ORIG)
out = [multi_gpu_test(det, items, depth=4, force_collective=True)["records"].clone() for _ in range(3)]
torch.cuda.synchronize()
head.use_graphs = backbone.use_graphs = False
bad = {0: [], 1: [], 2: []}
eager2_bad = []
for k, (im, m) in enumerate(items):
    r = head.simple_test_bboxes(backbone(im, slot=7), m)[0]
    sub, obj = head.pair_positions()
    want = pack_triplets(r[1], r[7], sub[0], obj[0]).clone()
    r = head.simple_test_bboxes(backbone(im, slot=6), m)[0]
    sub, obj = head.pair_positions()
    want2 = pack_triplets(r[1], r[7], sub[0], obj[0]).clone()
    if not torch.equal(want, want2):
        eager2_bad.append(k)
    for p in range(3):
        if not torch.equal(out[p][k].to(dev), want):
            d = (out[p][k].to(dev) - want).abs()
            bad[p].append((k, tuple(im.shape[-2:]), float(d.max()), int((d > 0).sum())))
print(arith, "passes equal:", torch.equal(out[0], out[1]), torch.equal(out[1], out[2]))
print("eager twice differs:", eager2_bad)
for p in range(3):
    print("pass", p, "differs from eager:", bad[p])
