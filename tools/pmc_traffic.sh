#!/bin/bash
# HBM traffic per kernel from rocprofv3 PMC counters (run on the GPU box through gpurun):
#   FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one pass), counters only
#   (no trace domains), on a short eager single-stream bench run.
# Output: gpurun_out/pmc/{fetch,write}/... csv + gpurun_out/pmc_traffic.json
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 2 --no-pipeline --no-graphs --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- $CMD > "$OUT/write.log" 2>&1
# matrix-pipe utilisation: SQ counters only, their own pass
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
  --output-format csv -d "$OUT/mfma" -o mfma -- $CMD > "$OUT/mfma.log" 2>&1
python "$ROOT/tools/pmc_traffic.py" "$OUT" > "$ROOT/gpurun_out/pmc_traffic.json"
ls -R "$OUT" | head -20
