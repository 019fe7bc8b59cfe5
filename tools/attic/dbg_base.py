import sys, torch
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests")
from helpers import baseline_cfg
from pairnet_amd import CrossHeadBaseline, CrossHead2, pairnet_head_cfg
from oracle import seeded
DEV="cuda:0"
for cls_,cfgf in ((CrossHeadBaseline, baseline_cfg),):
    from helpers import golden, oracle_baseline_head, overrides_of
    fx=golden("baseline_small"); _,sd,_=oracle_baseline_head(int(fx["weight_seed"]),overrides_of(fx))
    cfg=cfgf(); head=cls_(**cfg); head.load_state_dict(sd); head.return_all_layers=False; head.to(DEV)
    feats=[f.to(DEV) for f in seeded.seeded_feats(71,2,96,128)]
    metas=[dict(img_shape=(96,128,3),scale_factor=[2.0]*4)]*2
    outs={}
    feats2=[f*(1+2e-7) for f in feats]
    for fuse in (False,True,"perturbed"):
        head.fuse_chains=bool(fuse is True)
        c,m=head.forward(feats2 if fuse=="perturbed" else feats,metas)
        torch.cuda.synchronize()
        pl=head._last_plan
        outs[fuse]={k:v.clone() for k,v in dict(q=pl.q,qn=pl.qn,me=pl.me,cls=pl.cls,q1=pl.q1,q2=pl.q2,Qp=pl.Qp, MP=pl.MP).items()}
    for k in outs[True]:
        print(k, "fused-vs-unfused", float((outs[True][k]-outs[False][k]).abs().max()), " unfused on inputs scaled by (1+2e-7) vs unfused", float((outs["perturbed"][k]-outs[False][k]).abs().max()))
