#!/bin/bash
# One gpurun call that regenerates the judged profile set (run on the GPU box):
#   gpurun_out/<tag>_bench.json                  default `python bench.py` line (image -> triplets)
#   gpurun_out/<tag>_kernel_stats.csv            rocprofv3 --kernel-trace --stats of
#                                                `bench.py --no-extras --no-cpu-baseline`
#   gpurun_out/<tag>_kernel_stats_head_only.csv  same with `--path head`
#   (Swin-L / 200 queries and the box trunk are child legs of the default bench line since round 6)
#   gpurun_out/<tag>_train_step_kernel_stats.csv tools/train_step_probe.py under rocprofv3 (plain: bench line's `training_step`)
#   gpurun_out/pmc_traffic.json                  tools/pmc_traffic.sh (separate --pmc passes)
# usage: tools/profile_round.sh r04_v1
set -u
TAG="${1:-r04_vX}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 python "$ROOT/bench.py" 2> "$OUT/${TAG}_bench.err" | tail -1 > "$OUT/${TAG}_bench.json"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}" -o full -- \
  python "$ROOT/bench.py" --no-extras --no-cpu-baseline > "$OUT/prof_${TAG}_full.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}" -o head -- \
  python "$ROOT/bench.py" --path head --no-extras --no-cpu-baseline > "$OUT/prof_${TAG}_head.log" 2>&1
f=$(find "$OUT/prof_${TAG}" -name "full_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${TAG}_kernel_stats.csv"
f=$(find "$OUT/prof_${TAG}" -name "head_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${TAG}_kernel_stats_head_only.csv"
# the training step: phases + per-kernel view of tools/train_step_probe.py
# (its plain timings are the bench line's `training_step`)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}" -o train -- \
  python "$ROOT/tools/train_step_probe.py" 10 > "$OUT/prof_${TAG}_train.log" 2>&1
f=$(find "$OUT/prof_${TAG}" -name "train_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${TAG}_train_step_kernel_stats.csv"
# drop the bulky traces, keep the summaries
find "$OUT/prof_${TAG}" -name "*_kernel_trace.csv" -delete
bash "$ROOT/tools/pmc_traffic.sh" > "$OUT/pmc_${TAG}.log" 2>&1
ls -la "$OUT" | grep "$TAG"
