"""Tuning aid: sustained shader clock while the fp32 GEMM runs (rocm-smi sampled from a
second process) -- the 157.3 TFLOP/s roof assumes the 2.4 GHz peak clock."""
import os, subprocess, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev = "cuda:0"
M, N, K = 21950, 1024, 256
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1; o = torch.empty(M, N, device=dev)
samples = []
def sample():
    for _ in range(8):
        time.sleep(0.4)
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        samples.append([l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l])
th = threading.Thread(target=sample); th.start()
t0 = time.time(); n = 0
torch.cuda.synchronize()
while time.time() - t0 < 4.0:
    for _ in range(200): hip.linear(x, w, None, o)
    torch.cuda.synchronize(); n += 200
dt = time.time() - t0
th.join()
print("sustained %.1f TFLOP/s over %.1f s" % (2.0 * M * N * K * n / dt / 1e12, dt))
for s in samples: print(s)
