import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from test_losses_gpu import _outputs
from pairnet_amd import TailTrainer, ResNet50Hip
DEV="cuda:0"
for scope in ("tail", "all"):
    head, cls, masks, metas, gt_rels, gt_labels, gt_masks, pts = _outputs(2, H=96, W=128, bs=2)
    g = torch.Generator().manual_seed(2)
    feats = [torch.randn(2, c, 96 // s, 128 // s, generator=g).to(DEV) for c, s in zip((256, 512, 1024, 2048), (4, 8, 16, 32))]
    bb = None
    if scope == "all":
        bb = ResNet50Hip().to(DEV); feats = torch.randn(2, 3, 96, 128, generator=g).to(DEV)
    tr = TailTrainer(head, lr=float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4, train_decoder=scope=="all", backbone=bb)
    hist = []
    for i in range(150):
        out = tr.step(feats, metas, gt_rels, gt_labels, gt_masks, point_coords=pts)
        if i % 10 == 0 or i == 149:
            hist.append((i, round(float(out["loss_match"]),3), round(float(out["loss_r_cls"]),3), round(float(out["grad_norm"]),2)))
    print(scope, hist)
