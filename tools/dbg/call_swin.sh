#!/bin/sh
python -m pytest tests/test_gemm_s3_gpu.py tests/test_swin_gpu.py -x -q -m gpu 2>&1 | tail -12
for a in bf16x3 fp32; do
  python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 6 --queries 200 --in-channels 192,384,768,1536 --gemm-arithmetic $a 2>gpurun_out/swin_err.txt | tail -1 > gpurun_out/swin_$a.json
  tail -c 300 gpurun_out/swin_err.txt
  python - "$a" <<'PY'
import json, sys
d = json.load(open("gpurun_out/swin_%s.json" % sys.argv[1]))
print("swin-l", sys.argv[1], d["value"], d["ms_per_step"], str(d["pipeline_check"])[:50])
for k, v in list(d["kernel_profile"].items())[:9]:
    print("   ", k, round(v["ms_per_step"], 3), v["launches_per_step"], round(v["tflops"], 1))
PY
done
