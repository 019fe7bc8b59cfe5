"""How far is the GPU's importance matrix from the reference's, and how many of the
top-k positions agree exactly?  (design input for the tie-free fixtures)

    python tools/imp_noise_probe.py [fixture ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import golden, head_cfg, oracle_head, overrides_of  # noqa: E402
from oracle import seeded  # noqa: E402
from pairnet_amd import CrossHead2  # noqa: E402

DEV = "cuda:0"


def stats(tag, ref_imp, got_imp, ref_idx, got_idx):
    ref_imp, got_imp = np.asarray(ref_imp, np.float64), np.asarray(got_imp, np.float64)
    for b in range(ref_imp.shape[0]):
        r, g = ref_imp[b].reshape(-1), got_imp[b].reshape(-1)
        order = np.argsort(-r)[:101]
        gaps = r[order][:-1] - r[order][1:]
        print("%s[%d]: std %.3g  max err %.3g  err@top101 %.3g  min gap %.3g  exact %d/%d"
              % (tag, b, r.std(), np.abs(r - g).max(), np.abs(r - g)[order].max(), gaps.min(),
                 int((np.asarray(ref_idx[b]) == np.asarray(got_idx[b])).sum()), len(ref_idx[b])))


def run_e2e(name):
    fx = golden(name)
    _, sd, _ = oracle_head(int(fx["weight_seed"]), overrides_of(fx))
    H, W = int(fx["height"]), int(fx["width"])
    bs = int(fx["batch"]) if "batch" in fx.files else 1
    sf = 2.0 if name == "e2e_small" else 2.083
    feats = seeded.seeded_feats(int(fx["feat_seed"]), bs, H, W)
    metas = [dict(img_shape=(H, W, 3), scale_factor=[sf] * 4)] * bs
    head = CrossHead2(**head_cfg())
    head.load_state_dict(sd)
    head.to(DEV)
    for exact in (False, True):
        head.exact_mask_order = exact
        cls, _ = head.forward([f.to(DEV) for f in feats], metas)
        torch.cuda.synchronize()
        key = "cls_importance" if "cls_importance" in fx.files else "importance"
        stats("%s exact_mask_order=%s" % (name, exact), fx[key], cls["importance"].cpu().numpy(),
              fx["topk_idx"], head._last_plan.topk_idx.cpu().numpy())
        for k in ("cls", "rel"):
            kk = "cls_" + k if "cls_" + k in fx.files else k
            print("   %s err %.3g" % (k, float(np.abs(fx[kk] - cls[k].cpu().numpy()).max())))


def run_ppn():
    fx = golden("ppn")
    _, sd, _ = oracle_head(int(fx["weight_seed"]))
    head = CrossHead2(**head_cfg())
    head.load_state_dict(sd)
    head.to(DEV)
    pl = head._plan(1, [(3, 4), (6, 8), (12, 16)], (24, 32))
    pl.q.copy_(torch.from_numpy(fx["query_feat"]).to(DEV).transpose(0, 1).reshape(-1, 256))
    pl.cls.zero_()
    pl.MP.zero_()
    head._relation_stage(pl)
    torch.cuda.synchronize()
    print("ppn raw err %.3g" % float(np.abs(fx["importance_raw"] - pl.imp_raw.cpu().numpy()).max()))
    stats("ppn", fx["importance"], pl.imp.cpu().numpy(), fx["topk_idx"], pl.topk_idx.cpu().numpy())


if __name__ == "__main__":
    names = sys.argv[1:] or ["ppn", "e2e_small", "e2e_full"]
    for n in names:
        run_ppn() if n == "ppn" else run_e2e(n)
