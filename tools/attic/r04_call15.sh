cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/c15_full.log 2>&1
bash tools/profile_round.sh r04_v4 > $O/c15_profile.log 2>&1
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/c15_b20.err | tail -1 > $O/r04_v4_bench_steps20.json
timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 5000 2> $O/c15_sus.err | tail -1 > $O/r04_v4_sustained.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/c15_smoke.log 2>&1
tail -3 $O/c15_full.log; tail -2 $O/c15_smoke.log; python -c "
import json
for f in ('r04_v4_bench', 'r04_v4_bench_steps20', 'r04_v4_sustained', 'r04_v4_swinl_bench', 'r04_v4_bbox_bench'):
    d = json.load(open('$O/%s.json' % f)); print(f, round(d['value'],2), round(d['ms_per_step'],4), d['steps'], round(d['roofline']['frac'],4) if 'roofline' in d else None, (d.get('roofline_attention') or {}).get('avg_launch_us'))
"
