cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "linear_residual or msda" 2>&1 | tail -15) > $O/c2_kernels.log 2>&1
(timeout 300 python tools/msda_ab.py) > $O/c2_msda_ab.log 2>&1
(timeout 300 python tools/gemm_ln_probe.py) > $O/c2_gemm_ln.log 2>&1
(timeout 900 python -m pytest tests/test_dist.py -m gpu -x -q -k "multi_gpu_test" 2>&1 | tail -25) > $O/c2_dist.log 2>&1
(timeout 900 python -m pytest tests/test_production_gpu.py -x -q -s -k "r50_800x1333_image_to_triplets or graph_replay" 2>&1 | grep -v "^$" | tail -40) > $O/c2_prod.log 2>&1
for v in none proj proj,ffn none proj,ffn; do
  timeout 400 python bench.py --no-cpu-baseline --no-extras --enc-fused-ln $v 2> $O/c2_bench_$v.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d.get('pipeline_check'))" >> $O/c2_bench_ab.log 2>&1
done
cd /tmp; timeout 700 python $GRAFT_REPO_ROOT/tools/gemm_stalls.py --pass=8 --pass=9 --pass=10 > $GRAFT_REPO_ROOT/$O/c2_stalls.log 2>&1
cd $GRAFT_REPO_ROOT
tail -5 $O/c2_kernels.log; cat $O/c2_msda_ab.log $O/c2_gemm_ln.log; tail -8 $O/c2_dist.log; cat $O/c2_bench_ab.log; grep -E "mismatch|passed|failed|error" $O/c2_prod.log | tail
