"""Pipelined vs single-stream results with the split form on in the backbone / the head / both,
pipeline depth and stage-A stream count varied (no graphs)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import golden
import test_production_gpu as T
from pairnet_amd import PipelinedHead, hip
if os.environ.get('TRACE'):
    hip.SPLIT_TRACE = {}
DEV = "cuda:0"
fx = golden("e2e_image_full")
det = T._detector(fx, "r50")
img, metas = T._image(fx)
imgs = [img[i:i + 1].contiguous().to(DEV) for i in range(2)]
head, net = det.bbox_head, det.backbone
CASES = eval(os.environ.get('CASES', 'None')) or ((True, False, 4, 2), (False, True, 4, 2), (True, True, 4, 2), (True, True, 2, 1), (True, True, 4, 1), (False, False, 4, 2))
for bb, hd, depth, na in CASES:
    net.split_gemm, head.split_gemm = bb, hd
    eager = []
    for im in imgs:
        r = head.simple_test_bboxes(net(im), metas[:1])[0]
        eager.append([t.clone() for t in (r[1], r[7], r[4])])
    pipe = PipelinedHead(head, depth=depth, a_streams=na)
    order = [0, 1, 1, 0, 0, 1, 0, 1, 1, 0, 1, 0] * 2
    got = []
    for i in order:
        sl = pipe.count % len(pipe.streams_a)
        pipe.streams_a[sl].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(pipe.streams_a[sl]):
            res = pipe.submit(net(imgs[i], slot=sl), metas[:1])
            if res is not None:
                got.append([t.clone() for t in (res[0][1], res[0][7], res[0][4])])
    while pipe.queue:
        res = pipe._finish(pipe.queue.pop(0))
        got.append([t.clone() for t in (res[0][1], res[0][7], res[0][4])])
    torch.cuda.synchronize()
    bad = [n for n, (i, g) in enumerate(zip(order, got)) if not all(bool(torch.equal(a, b)) for a, b in zip(eager[i], g))]
    for n in bad[:4]:
        i = order[n]
        print("    image", n, "max |d r_dists| %.3e" % (got[n][1] - eager[i][1]).abs().max().item(),
              "labels differing", int((got[n][0] != eager[i][0]).sum()), "pan pixels differing", int((got[n][2] != eager[i][2]).sum()))
    if hip.SPLIT_TRACE is not None:
        for k, v in sorted(hip.SPLIT_TRACE.items()):
            if v[1].item():
                print("    GEMM", k, "launches", int(v[0].item()), "re-run differs", int(v[1].item()))
        print("    traced shapes:", len(hip.SPLIT_TRACE))
        hip.SPLIT_TRACE.clear()
    print("backbone split", bb, "head split", hd, "depth", depth, "a_streams", na, "-> mismatching images", len(bad), "of", len(order), bad[:8], flush=True)
