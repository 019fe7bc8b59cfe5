"""Who is the victim?  (A) an image through the DEFAULT kernels while split GEMMs hammer another
stream; (B) split GEMMs while an image runs through the default kernels on another stream."""
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
im = img[0:1].contiguous().to(DEV)
head, net = det.bbox_head, det.backbone
net.split_gemm = head.split_gemm = False
def image():
    f = net(im)
    r = head.simple_test_bboxes(f, metas[:1])[0]
    return [t.clone() for t in f] + [r[1].clone(), r[7].clone(), r[4].clone(), head._last_plan.X.clone(), head._last_plan.MF.clone()]
names = ["c2", "c3", "c4", "c5", "labels", "r_dists", "pan", "X", "MF"]
want = image(); torch.cuda.synchronize()
M, N, K = 21950, 1024, 256
x1, w1 = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / 16
wantg = torch.empty(M, N, device=DEV)
hip.gemm(x1, w1, wantg, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, split=True)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for mode in ["split"] * 12 + ["default"] * 3:
    outs = [torch.empty(M, N, device=DEV) for _ in range(60)]
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        for o in outs:
            hip.gemm(x1, w1, o, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, split=(mode == "split"))
    with torch.cuda.stream(s2):
        got = image()
    torch.cuda.synchronize()
    if all(bool(torch.equal(a, b)) for a, b in zip(want, got)) and (mode != "split" or all(torch.equal(o, wantg) for o in outs)):
        print("GEMM loop:", mode, "all equal", flush=True)
        continue
    print("GEMM loop:", mode, "| image buffers equal:", dict(zip(names, [bool(torch.equal(a, b)) for a, b in zip(want, got)])),
          "| wrong GEMM outputs:", sum(not torch.equal(o, wantg) for o in outs) if mode == "split" else "-", flush=True)
