"""Feasibility probe: CU-masked HIP streams (hipExtStreamCreateWithCUMask) under torch."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip as ph
hiprt = ctypes.CDLL("libamdhip64.so")
def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hiprt.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
dev = "cuda:0"
M, N, K = 21950, 1024, 256
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1; o = torch.empty(M, N, device=dev)
def T(stream, n=10):
    with torch.cuda.stream(stream):
        for _ in range(3): ph.linear(x, w, None, o)
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): ph.linear(x, w, None, o)
        e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / n
print("unmasked stream: %.1f us" % T(torch.cuda.Stream()))
full = [0xFFFFFFFF] * 8
print("all 256 bits   : %.1f us" % T(masked_stream(full)))
def rng(a, b):
    words = [0] * 8
    for i in range(a, b):
        words[i // 32] |= 1 << (i % 32)
    return words
for a, b in ((0, 8), (0, 16), (0, 32), (0, 64), (0, 128), (0, 192), (0, 224), (0, 240), (0, 248),
             (16, 256), (32, 256), (8, 256), (128, 256), (240, 256), (64, 128)):
    print("bits [%3d,%3d) (%3d set): %8.1f us" % (a, b, b - a, T(masked_stream(rng(a, b)))))
