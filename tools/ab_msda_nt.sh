#!/bin/bash
# k_msda with non-temporal offset / logit loads and output stores (-DMSDA_NT=1 variant library)
# against the product: stand-alone times, HBM traffic of the probe (FETCH_SIZE / WRITE_SIZE in
# separate counter passes), alternating pipelined benches.
cd "$(dirname "$0")/.."
ROOT=$(pwd); mkdir -p gpurun_out/msda_nt
{
echo "== product"; python tools/msda_ab.py 2>&1 | grep -v amdgpu.ids | grep "one-shot 8"
echo "== nt variant"; LIB=tools/bin/libpn_msda_nt.so python tools/msda_ab.py 2>&1 | grep -v amdgpu.ids | grep "one-shot 8"
cd /tmp && export TMPDIR=/tmp
for v in product nt; do
  L=""; [ $v = nt ] && L="$ROOT/tools/bin/libpn_msda_nt.so"
  for c in FETCH_SIZE WRITE_SIZE; do
    LIB=$L timeout 300 rocprofv3 --pmc $c --output-format csv -d $ROOT/gpurun_out/msda_nt/${v}_$c -o p -- python $ROOT/tools/msda_ab.py > /dev/null 2>&1
    f=$(find $ROOT/gpurun_out/msda_nt/${v}_$c -name "*counter_collection.csv" | head -1)
    python - "$f" "$v $c" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Kernel_Name"].startswith("void k_msda<3>") or r["Kernel_Name"].startswith("k_msda<3>"):
        tot[r["Counter_Name"]][0] += float(r["Counter_Value"]); tot[r["Counter_Name"]][1] += 1
for k, (v, n) in tot.items():
    print(sys.argv[2], k, "per launch (raw counter units x 1):", v / max(n, 1), "launches", n)
PY
  done
done
cd $ROOT
one() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "== bench product $i"; python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | one
  echo "== bench nt $i"; python tools/bench_variant.py tools/bin/libpn_msda_nt.so --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | one
done
} > gpurun_out/ab_msda_nt.txt 2>&1
cat gpurun_out/ab_msda_nt.txt
