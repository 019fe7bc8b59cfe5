#!/bin/sh
python -m pytest tests/test_backbone_gpu.py -x -q -m gpu 2>&1 | tail -6
for v in 0 64 128 256; do
  python bench.py --no-cpu-baseline --no-extras --bb-set s3_conv3_min_planes=$v 2>gpurun_out/bb_err.txt | tail -1 > gpurun_out/bb_$v.json
  tail -c 200 gpurun_out/bb_err.txt
  python - "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bb_%s.json" % sys.argv[1]))
print("s3_conv3_min_planes", sys.argv[1], d["value"], d["ms_per_step"], str(d["pipeline_check"])[:40])
PY
done
