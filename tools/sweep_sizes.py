"""Robustness sweep (not a test): both heads against their CPU oracles at odd image sizes and
batch sizes, seeded weights, reporting the worst output errors.  Index selections are compared
as sets with a near-tie allowance (random weights: see LABNOTES.md section 3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from collections import OrderedDict
import numpy as np, torch
from oracle import seeded
import pairnet_amd as P
from helpers import oracle_head

DEV = "cuda:0"
E = lambda a, b: float((a.detach().cpu().double() - b.double()).abs().max())


def sweep_pairnet(sizes):
    ohead, sd, _ = oracle_head(777)
    cfg = P.pairnet_head_cfg(); cfg.pop("type")
    head = P.CrossHead2(**cfg).to(DEV); head.load_state_dict(sd)
    for bs, H, W in sizes:
        feats = seeded.seeded_feats(1000 + H, bs, H, W)
        metas = [dict(img_shape=(H, W, 3), scale_factor=[1.7] * 4)] * bs
        with torch.no_grad():
            oc, om = ohead.forward(feats, metas)
        hc, hm = head.forward([f.to(DEV) for f in feats], metas)
        same = float((hc["importance"].cpu().flatten(1).topk(100)[1].sort(-1)[0] ==
                      oc["importance"].flatten(1).topk(100)[1].sort(-1)[0]).float().mean())
        print("CrossHead2 bs=%d %dx%d: cls %.1e importance %.1e mask %.1e rel %.1e | top-k set overlap %.2f"
              % (bs, H, W, E(hc["cls"], oc["cls"]), E(hc["importance"], oc["importance"]),
                 E(hm["mask"], om["mask"]) / float(om["mask"].abs().max()), E(hc["rel"], oc["rel"]), same), flush=True)


def sweep_bbox(sizes):
    from oracle.bbox_head import OracleCrossHeadBBox
    from oracle.deformable_detr import ChannelMapper
    cfg = {k: v for k, v in P.bbox_head_cfg().items() if k != "type"}
    ncfg = {k: v for k, v in P.channel_mapper_cfg().items() if k != "type"}
    oh, on = OracleCrossHeadBBox(**cfg).eval(), ChannelMapper(**ncfg).eval()
    sd = seeded.seeded_state_dict(OrderedDict((k, tuple(v.shape)) for k, v in oh.state_dict().items()), 31)
    nsd = seeded.seeded_state_dict(OrderedDict((k, tuple(v.shape)) for k, v in on.state_dict().items()), 32)
    oh.load_state_dict(sd); on.load_state_dict(nsd)
    hh = P.CrossHeadBBox(**cfg).to(DEV); hh.load_state_dict(sd)
    hn = P.ChannelMapper(**ncfg).to(DEV); hn.load_state_dict(nsd)
    for bs, H, W in sizes:
        feats = seeded.smooth_feats(2000 + H, bs, H, W, 8)[1:]
        metas = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=[1.3] * 4)] * bs
        tr = {}
        with torch.no_grad():
            oc, ob = oh(on(feats), metas, trace=tr)
        hc, hb = hh(hn([f.to(DEV) for f in feats]), metas)
        pl = hh._last_plan
        pset = np.mean([len(set(pl.top_idx[i].tolist()) & set(tr["topk_proposals"][i].tolist())) / 300 for i in range(bs)])
        # proposals tied at the constant score of the zeroed (invalid) tokens may be picked in
        # any order (torch.topk leaves it unspecified); the tied tokens give IDENTICAL queries,
        # so what must agree is the multiset of query scores and of refined boxes
        valid = P.CrossHeadBBox.proposals(pl.shapes)[1]
        n_inv = int((~valid[tr["topk_proposals"][0]]).sum())
        qs = E(pl.qscore.sort(-1)[0], tr["query_score"].sort(-1)[0])
        bx = E(pl.ref[-1].view(bs, -1, 4).sum(-1).sort(-1)[0], tr["coords"][-1].sum(-1).sort(-1)[0])
        print("CrossHeadBBox bs=%d %dx%d (%d tokens): memory %.1e enc_cls %.1e enc_box %.1e | proposal "
              "overlap %.3f (invalid tokens among the reference's proposals: %d) | sorted query scores "
              "%.1e sorted box sums %.1e"
              % (bs, H, W, pl.SN, E(pl.X, tr["memory"]), E(hc["enc_cls_scores"], oc["enc_cls_scores"]),
                 E(hc["enc_bbox_preds"], oc["enc_bbox_preds"]), pset, n_inv, qs, bx), flush=True)


if __name__ == "__main__":
    sweep_bbox([(1, 203, 317), (3, 160, 255), (1, 608, 1023), (2, 480, 641)])
    if len(sys.argv) < 2:
        sweep_pairnet([(1, 203, 317), (3, 97, 131), (1, 608, 1023), (2, 480, 641)])
