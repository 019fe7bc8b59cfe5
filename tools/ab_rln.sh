#!/bin/bash
# A/B of k_gemm_rowln's LDS image: swizzled (product, 4 workgroups per CU) against padded
# (tools/bin/libpn_rln_pad.so, 3 per CU): stand-alone probe, bitwise test, alternating benches.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "linear_residual" 2>&1 | tail -2
echo "== swizzled (product)"; python tools/gemm_ln_probe.py
echo "== padded (variant)"; LIB=tools/bin/libpn_rln_pad.so python tools/gemm_ln_probe.py
for i in 1 2 3; do
  echo "== bench swizzled $i"; python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  echo "== bench padded $i"; python tools/bench_variant.py tools/bin/libpn_rln_pad.so --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
} > gpurun_out/ab_rln.txt 2>&1
cat gpurun_out/ab_rln.txt
