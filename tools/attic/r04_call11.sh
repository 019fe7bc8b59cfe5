cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/c11_full.log 2>&1
bash tools/profile_round.sh r04_v3 > $O/c11_profile.log 2>&1
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/c11_b20.err | tail -1 > $O/r04_v3_bench_steps20.json
timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 5000 2> $O/c11_sus.err | tail -1 > $O/r04_v3_sustained.json
tail -4 $O/c11_full.log; python -c "
import json
for f in ('r04_v3_bench', 'r04_v3_bench_steps20', 'r04_v3_sustained', 'r04_v3_swinl_bench', 'r04_v3_bbox_bench'):
    d = json.load(open('$O/%s.json' % f)); print(f, round(d['value'],2), round(d['ms_per_step'],4), d['steps'], round(d['roofline']['frac'],4) if 'roofline' in d else None, (d.get('roofline_deformable_sampling') or {}).get('avg_launch_us'))
"
