"""Two whole image flows (backbone + head, eager) on two streams at once, separate slots, split
form on (PAIRNET_SPLIT_GEMM=1) -- against the same flows run one after the other."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import golden
import test_production_gpu as T
from pairnet_amd import hip
DEV = "cuda:0"
fx = golden("e2e_image_full")
det = T._detector(fx, "r50")
img, metas = T._image(fx)
imgs = [img[i:i + 1].contiguous().to(DEV) for i in range(2)]
head, net = det.bbox_head, det.backbone
names = ["c2", "c3", "c4", "c5", "X", "MF", "VOA", "q", "r_dists"]
def flow(i, slot):
    f = net(imgs[i], slot=slot)
    B, shapes, hw2 = head._check_feats(f, metas[:1])
    pl = head._plan(B, shapes, hw2, slot, head._feats_nhwc)
    head._run_stage("a", pl, f)
    head._run_stage("b", pl)
    return [t.clone() for t in f] + [pl.X.clone(), pl.MF.clone(), pl.VOA.clone(), pl.q.clone(), pl.rel.clone()]
want = [flow(0, 0), flow(1, 1)]
torch.cuda.synchronize()
ss = [torch.cuda.Stream(), torch.cuda.Stream()]
for rep in range(6):
    got = [None, None]
    torch.cuda.synchronize()
    for j in range(2):
        with torch.cuda.stream(ss[j]):
            got[j] = flow(j, j)
    torch.cuda.synchronize()
    for j in range(2):
        eq = [bool(torch.equal(a, b)) for a, b in zip(want[j], got[j])]
        if not all(eq):
            print("rep", rep, "flow", j, dict(zip(names, eq)),
                  {n: "%.2e" % (a - b).abs().max().item() for n, a, b in zip(names, want[j], got[j]) if not torch.equal(a, b)}, flush=True)
print("done")
