cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/c22_full.log 2>&1
bash tools/profile_round.sh r04_v5 > $O/c22_profile.log 2>&1
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/c22_b20.err | tail -1 > $O/r04_v5_bench_steps20.json
timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 5000 2> $O/c22_sus.err | tail -1 > $O/r04_v5_sustained.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/c22_smoke.log 2>&1
tail -3 $O/c22_full.log; tail -2 $O/c22_smoke.log; python -c "
import json
for f in ('r04_v5_bench', 'r04_v5_bench_steps20', 'r04_v5_sustained', 'r04_v5_swinl_bench', 'r04_v5_bbox_bench'):
    d = json.load(open('$O/%s.json' % f)); r = d.get('roofline', {}); print(f, round(d['value'],2), round(d['ms_per_step'],4), d['steps'], round(r.get('frac', 0),4), round(r.get('frac_of_sustained_clock_roof', 0),4))
"
