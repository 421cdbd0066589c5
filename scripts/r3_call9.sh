#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 900 python bench.py > $O/r3c9_bench.json 2> $O/r3c9_bench.err; tail -2 $O/r3c9_bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3c9_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['other_path'], d['parity_setting'], d['u_err'], d['cpu_baseline']['value'], d['cold'], d['hbm_leg'], json.dumps(d['latency'])[:900], d['real_osqp'])
PY
for B in 128 256 512; do timeout 300 python bench.py --batch $B --no-cpu-baseline --no-other-path --no-refactor-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B', d['config']['batch_per_gpu'], round(d['value']), round(d['ms_per_step'],4), d['roofline']['kernel'], d['roofline']['frac'])"; done
timeout 600 python bench.py --workload cfg5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5', round(d['value']), round(d['ms_per_step'],4), d['roofline']['frac'], d['other_path'], d['parity_setting'])"
