#!/bin/sh
# A/B of S3_LATE_B on the LN shapes (96-row kernel): correctness test + rates per variant
for v in 0 2 3 4 6; do
  export PAIRNET_LIB=$PWD/tools/gpubin/libpairnet_late$v.so
  echo "== S3_LATE_B=$v"
  python -m pytest tests/test_gemm_s3_gpu.py -x -q -m gpu -k "integer or layernorm" 2>&1 | tail -1
  python tools/gemm_s3_bench.py 2>&1 | grep -A4 "N=256 K=1024\|N=256 K=256" | grep "M=\|s3 LN -> S3\|s3 -> fp32$"
done
