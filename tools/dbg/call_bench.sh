#!/bin/sh
# one GPU call: tests given as $1 (pytest selection), the full bench line + a readable digest
python -m pytest $1 -x -q -m gpu 2>&1 | tail -4
python bench.py --no-cpu-baseline $2 2>gpurun_out/bench_err.txt | tail -1 > gpurun_out/bench_latest.json
tail -c 400 gpurun_out/bench_err.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_latest.json"))
print("HEADLINE", d["value"], d["ms_per_step"], "| fp32 encoder:", (d.get("fp32_mfma_encoder") or {}).get("images_per_s"))
r = d.get("roofline") or {}
print("dominant", r.get("kernel"), r.get("frac"), r.get("ms_per_step"))
for k in ("roofline_k_gemm_s3", "roofline_k_gemm_s3_ln"):
    r = d.get(k) or {}
    print(k, r.get("achieved"), r.get("frac"), r.get("fp32_equivalent_tflops"), r.get("ms_per_step"), r.get("launches_per_step"))
print(d.get("roofline_all_gemm_kernels"))
for k in ("swin_l_200q", "box_trunk"):
    v = d.get(k)
    print(k, v if not isinstance(v, dict) else (v.get("value"), v.get("ms_per_step")))
for k, v in list(d.get("kernel_profile", {}).items())[:14]:
    print("  ", k, round(v["ms_per_step"], 3), v["launches_per_step"])
print("latency", d.get("latency_ms_single_stream_graphs"), "shape mix", d.get("shape_mix_product_loop", {}).get("images_per_s"),
      d.get("shape_mix_product_loop", {}).get("second_pass_images_per_s"))
PY
