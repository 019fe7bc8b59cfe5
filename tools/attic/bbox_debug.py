"""Stage-by-stage comparison of the HIP CrossHeadBBox with the CPU oracle (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from collections import OrderedDict
from oracle import seeded
from oracle.bbox_head import OracleCrossHeadBBox
from oracle.deformable_detr import ChannelMapper as OracleMapper
import pairnet_amd as P

cfg = {k: v for k, v in P.bbox_head_cfg().items() if k != "type"}
ncfg = {k: v for k, v in P.channel_mapper_cfg().items() if k != "type"}
oh, on = OracleCrossHeadBBox(**cfg).eval(), OracleMapper(**ncfg).eval()
sd = seeded.seeded_state_dict(OrderedDict((k, tuple(v.shape)) for k, v in oh.state_dict().items()), 12347)
nsd = seeded.seeded_state_dict(OrderedDict((k, tuple(v.shape)) for k, v in on.state_dict().items()), 12348)
oh.load_state_dict(sd); on.load_state_dict(nsd)
H, W, bs = 160, 192, 2
metas = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=[1.5, 1.25, 1.5, 1.25])] * bs
feats = seeded.seeded_feats(int(sys.argv[1]) if len(sys.argv) > 1 else 100, bs, H, W)[1:]
tr = {}
with torch.no_grad():
    nf = on(feats)
    oc, ob = oh(nf, metas, trace=tr)
    ores = oh.get_bboxes(oc, ob, metas, rescale=True)
dev = "cuda:0"
hn = P.ChannelMapper(**ncfg).to(dev); hn.load_state_dict(nsd)
hh = P.CrossHeadBBox(**cfg).to(dev); hh.load_state_dict(sd)
E = lambda a, b: float((a.detach().cpu().double() - b.double()).abs().max())
for fmt in ("nchw", "nhwc"):
    ins = [f.to(dev) if fmt == "nchw" else f.to(dev).contiguous(memory_format=torch.channels_last) for f in feats]
    outs = hn(ins)
    print("neck", fmt, [E(a, b) for a, b in zip(outs, nf)])
hc, hb = hh(outs, metas)
torch.cuda.synchronize()
pl = hh._last_plan
print("tokens in place:", not pl.own_tokens)
print("memory", E(pl.X, tr["memory"]), float(tr["memory"].abs().max()))
print("enc_cls", E(hc["enc_cls_scores"], oc["enc_cls_scores"]))
fin = torch.isfinite(torch.logit(oc["enc_bbox_preds"])).all(-1)
print("enc_box", E(hc["enc_bbox_preds"], oc["enc_bbox_preds"]))
a, b = pl.top_idx.cpu(), tr["topk_proposals"]
print("proposals: same set", [set(a[i].tolist()) == set(b[i].tolist()) for i in range(bs)], "same order", bool((a == b).all()))
print("query", E(pl.ptn[:, 256:].reshape(bs, -1, 256), tr["query"]), "query_pos", E(pl.ptn[:, :256].reshape(bs, -1, 256), tr["query_pos"]))
print("hs[-1]", E(pl.x.view(bs, -1, 256), tr["hs"][-1]))
print("classes", E(pl.classes, tr["classes"][-1]), "coords", E(pl.ref[-1].view(bs, -1, 4), tr["coords"][-1]))
print("query_score", E(pl.qscore, tr["query_score"]), float(tr["query_score"].max()))
print("keep same", bool((pl.keep.cpu() == tr["index"]).all()))
print("importance_raw", E(pl.imp_raw, tr["importance_raw"]), "topk same", bool((pl.topk_idx.cpu() == tr["topk_idx"]).all()))
for k in hc:
    print("cls", k, E(hc[k], oc[k]))
for k in hb:
    print("bbox", k, E(hb[k], ob[k]))
res = hh.get_bboxes(hc, hb, metas, rescale=True)
for r, o in zip(res, ores):
    print("get_bboxes", [E(x.float(), y.float()) for x, y in zip(r, o)])
