import sys, torch
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests")
from helpers import baseline_cfg, golden, oracle_baseline_head, overrides_of
from pairnet_amd import CrossHeadBaseline, hip
from oracle import seeded
DEV="cuda:0"
fx=golden("baseline_small"); _,sd,_=oracle_baseline_head(int(fx["weight_seed"]),overrides_of(fx))
head=CrossHeadBaseline(**baseline_cfg()); head.load_state_dict(sd); head.return_all_layers=False; head.to(DEV)
feats=[f.to(DEV) for f in seeded.seeded_feats(int(fx["feat_seed"]),2,96,128)]
metas=[dict(img_shape=(96,128,3),scale_factor=[2.0]*4)]*2
rec={}
orig_ffn, orig_attn = hip.ffn_ln, hip.attention
def run(fuse):
    log=[]
    def ffn(x,*a,**k):
        r=orig_ffn(x,*a,**k); log.append(("ffn_in",x.clone())); log.append(("ffn_out",a[6].clone())); return r
    def attn(q,ldq,k_,ldk,v,ldv,bits,rowall,out,ldo,scr,B,Q,Nk,scale):
        r=orig_attn(q,ldq,k_,ldk,v,ldv,bits,rowall,out,ldo,scr,B,Q,Nk,scale)
        log.append(("attn_q",q.clone() if q.is_contiguous() else q.contiguous().clone())); log.append(("attn_out",out.clone())); return r
    hip.ffn_ln, hip.attention = ffn, attn
    head.fuse_chains=fuse
    head.forward(feats,metas); torch.cuda.synchronize()
    hip.ffn_ln, hip.attention = orig_ffn, orig_attn
    return log
a=run(False); b=run(True)
print(len(a),len(b))
for i,((n1,x),(n2,y)) in enumerate(zip(a,b)):
    if x.shape==y.shape:
        print(i,n1,n2,float((x-y).abs().max()))
    else: print(i,n1,n2,"shape",x.shape,y.shape)
    pass
