"""Tuning aid: time the three pieces of the Winograd 3x3 convolution at 200x334x256."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev = "cuda:0"
B, H, W, C = 1, 200, 334, 256
x = torch.randn(B, H, W, C, device=dev); wt = torch.randn(256, 256, 3, 3, device=dev) * 0.02
U = hip.winograd_weights(wt); T = B * (H // 2) * (W // 2)
V = torch.empty(16, T, C, device=dev); Mb = torch.empty(16, T, C, device=dev); out = torch.empty(B, H, W, C, device=dev)
wp = wt.permute(0, 2, 3, 1).reshape(256, -1).contiguous()
def t(fn, n=10):
    for _ in range(3): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) * 1e3 / n
L = hip.lib()
print("input transform  %.1f us" % t(lambda: L.pn_winograd_f23_input_f32(x.data_ptr(), V.data_ptr(), B, H, W, C, None)))
print("batched GEMM     %.1f us" % t(lambda: hip.gemm(V, U, Mb, M=T, N=C, K=C, lda=C, ldw=C, ldc=C, batch=16, sA=T * C, sW=C * C, sC=T * C)))
print("output transform %.1f us" % t(lambda: L.pn_winograd_f23_output_f32(Mb.data_ptr(), None, out.data_ptr(), B, H, W, C, 0, None)))
print("whole winograd   %.1f us" % t(lambda: hip.conv3x3_winograd(x, U, None, out, V, Mb, B, H, W, C, C, False)))
print("direct conv      %.1f us" % t(lambda: hip.conv2d_nhwc(x, wp, None, out, B, H, W, C, C, 3, 3, 1, False)))
T4 = B * ((H + 3) // 4) * ((W + 3) // 4)
U4 = hip.winograd43_weights(wt); V4 = torch.empty(36, T4, C, device=dev); M4 = torch.empty(36, T4, C, device=dev)
out4 = torch.empty(B, H, W, C, device=dev)
print("F(4,3) input     %.1f us" % t(lambda: L.pn_winograd_f43_input_f32(x.data_ptr(), V4.data_ptr(), B, H, W, C, None)))
print("F(4,3) GEMM      %.1f us" % t(lambda: hip.gemm(V4, U4, M4, M=T4, N=C, K=C, lda=C, ldw=C, ldc=C, batch=36, sA=T4 * C, sW=C * C, sC=T4 * C)))
print("F(4,3) output    %.1f us" % t(lambda: L.pn_winograd_f43_output_f32(M4.data_ptr(), None, out4.data_ptr(), B, H, W, C, 0, None)))
print("F(4,3) whole     %.1f us" % t(lambda: hip.conv3x3_winograd43(x, U4, None, out4, V4, M4, B, H, W, C, C, False)))
hip.conv2d_nhwc(x, wp, None, out, B, H, W, C, C, 3, 3, 1, False)
ref = out.double(); sc = ref.abs().max().item()
hip.conv3x3_winograd(x, U, None, out, V, Mb, B, H, W, C, C, False)
print("F(2,3) vs direct: max |diff| / max|ref| = %.2e" % ((out.double() - ref).abs().max().item() / sc))
print("F(4,3) vs direct: max |diff| / max|ref| = %.2e" % ((out4.double() - ref).abs().max().item() / sc))
