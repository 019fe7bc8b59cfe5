"""Evidence for the "clock, not the kernel" reading of profiles/r04_gemm_stalls.json: socket
power and reported shader clock (rocm-smi, sampled from a thread every 0.25 s) while
  (1) the chip idles,
  (2) a REGISTER-ONLY fp32 MFMA loop runs (no LDS, no memory: tools/bin/mfma_power_probe if built),
  (3) the path's FFN-1 GEMM (21950 x 1024 x 256) runs back to back on N(0,1) operands,
  (4) the same GEMM on all-zero operands,
each for ~3 s, with the achieved TFLOP/s of (3) and (4).  Writes one JSON object to stdout."""
import json, os, re, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev = "cuda:0"
M, N, K = 21950, 1024, 256


def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = re.findall(r"sclk clock level:.*?\((\d+)Mhz\)", out)
    pw = re.findall(r"(?:Average|Current Socket) Graphics Package Power \(W\):\s*([\d.]+)", out)
    return (int(sclk[0]) if sclk else None, float(pw[0]) if pw else None)


def sampled(run, seconds=3.0):
    samples, stop = [], [False]

    def loop():
        while not stop[0]:
            samples.append(smi())
            time.sleep(0.25)
    th = threading.Thread(target=loop)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        n += run()
        torch.cuda.synchronize()
    dt = time.time() - t0
    stop[0] = True
    th.join()
    s = [x for x in samples[2:] if x[0] is not None]
    return dict(seconds=round(dt, 2), launches=n,
                sclk_mhz=[x[0] for x in s], power_w=[x[1] for x in s])


def gemm_phase(x, w, force=None):
    o = torch.empty(M, N, device=dev)

    def run():
        for _ in range(100):
            hip.linear(x, w, None, o, force=force)
        return 100
    r = sampled(run)
    r["tflops"] = round(2.0 * M * N * K * r["launches"] / r["seconds"] / 1e12, 1)
    return r


out = {"what": __doc__.strip().splitlines()[0]}
out["idle"] = sampled(lambda: (time.sleep(0.2), 0)[1], 1.5)
x = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.1
out["gemm_ffn1_normal_operands"] = gemm_phase(x, w)
out["gemm_ffn1_zero_operands"] = gemm_phase(torch.zeros_like(x), torch.zeros_like(w))
out["gemm_ffn1_tile128x64_normal"] = gemm_phase(x, w, "tile128x64")
out["gemm_ffn1_tile128x128_normal"] = gemm_phase(x, w, "tile")
out["gemm_ffn1_tile128x128_zero"] = gemm_phase(torch.zeros_like(x), torch.zeros_like(w), "tile")
probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "mfma_power_probe")
if os.path.exists(probe):
    th_out = {}

    def run_probe():      # (several back-to-back runs: one is only ~0.6 s of kernels)
        for _ in range(6):
            th_out["stdout"] = subprocess.run([probe], capture_output=True, text=True, timeout=60).stdout[-600:]
    th = threading.Thread(target=run_probe)
    th.start()
    time.sleep(0.5)
    s = []
    while th.is_alive():
        s.append(smi())
        time.sleep(0.25)
    th.join()
    out["register_only_mfma_loop"] = dict(sclk_mhz=[a for a, b in s if a], power_w=[b for a, b in s if a],
                                          stdout_tail=th_out.get("stdout"))
print(json.dumps(out, indent=1))
