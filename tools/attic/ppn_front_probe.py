import sys, torch; sys.path.insert(0, "/root/repo")
import importlib; hip = importlib.import_module("pair-net_amd.hip")
B, Q = 1, 100
se, oe = torch.randn(B, Q, 256, device="cuda"), torch.randn(B, Q, 256, device="cuda")
w1, b1 = torch.randn(64, 49, device="cuda"), torch.randn(64, device="cuda")
raw, c1 = torch.empty(B, Q, Q, device="cuda"), torch.empty(B, Q, Q, 64, device="cuda")
sn, on = torch.empty_like(se), torch.empty_like(oe)
def a(): hip.ppn_front(se, oe, w1, b1, raw, c1, B, Q)
def b():
    hip.l2normalize(se.view(-1, 256), sn.view(-1, 256)); hip.l2normalize(oe.view(-1, 256), on.view(-1, 256))
    hip.gemm(sn, on, raw, M=Q, N=Q, K=256, lda=256, ldw=256, ldc=Q, batch=B, sA=Q*256, sW=Q*256, sC=Q*Q)
    hip.mlearner_first(raw, w1, b1, c1, B, Q)
for name, f in (("fused", a), ("3-launch", b)):
    g = torch.cuda.CUDAGraph(); f(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20): f()
    for _ in range(3): g.replay()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize(); print(name, e0.elapsed_time(e1) / 400 * 1e3, "us")
