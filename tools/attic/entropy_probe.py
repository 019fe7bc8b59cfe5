"""Tuning aid: fp32 GEMM variants on N(0,1) data vs low-entropy data (values k/4, |k| <= 8):
how much of each tile shape's rate is taken by the power management when the operand buses
toggle (LABNOTES.md 6: the MFMA pipe alone sustains 154 TFLOP/s on N(0,1) register operands,
tools/mfma_power_probe.hip)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev = "cuda:0"
def T(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)*1e3/n
shapes = [(21950,1024,256),(21950,256,1024),(65536,1024,1024)]
for M,N,K in shapes:
    for what in ("normal", "lowent", "zeros"):
        if what == "normal":
            x=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)*0.1
        elif what == "lowent":
            x=(torch.randn(M,K,device=dev)*2).round().clamp(-8,8)/4; w=(torch.randn(N,K,device=dev)*2).round().clamp(-8,8)/4
        else:
            x=torch.zeros(M,K,device=dev); w=torch.zeros(N,K,device=dev)
        o=torch.empty(M,N,device=dev)
        row=[]
        for name,kw in (("tile64",dict(force=None)),("t128x64",dict(force="tile128x64")),("t128",dict(force="tile")),("split64",dict(split=True))):
            us=T(lambda: hip.linear(x,w,None,o,**kw))
            row.append("%s %6.1fus %5.1fTF" % (name, us, 2.0*M*N*K/us/1e6))
        print(M,N,K,"%-7s"%what," | ".join(row))
