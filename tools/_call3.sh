cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $O/c3_full.log 2>&1
(timeout 600 python -m pytest tests/test_production_gpu.py -x -q -s -k "swin_l_200 or streamer" 2>&1 | grep -E "mismatch|passed|failed|rror|min gap|C[2-5] " | tail -30) > $O/c3_prod.log 2>&1
(timeout 300 python -m pytest tests/test_losses_gpu.py -x -q -s 2>&1 | grep -E "loss values|passed|failed|rror" | tail -10) > $O/c3_loss.log 2>&1
(timeout 300 python tools/msda_ab.py) > $O/c3_msda_ab.log 2>&1
(timeout 200 python tools/gemm_ln_probe.py) > $O/c3_gemm_ln.log 2>&1
tail -25 $O/c3_full.log; cat $O/c3_prod.log $O/c3_loss.log $O/c3_msda_ab.log $O/c3_gemm_ln.log
