#!/bin/bash
# round-3 GPU call 1: regression with the dense backend on and off, the new bench line, low-occupancy baselines, phase split
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/r3c1_tests.txt
MPCQP_DENSE=0 timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/r3c1_tests_nodense.txt
timeout 900 python bench.py > $O/r3c1_bench.json 2> $O/r3c1_bench.err
timeout 300 python bench.py --workload cfg2 > $O/r3c1_cfg2_dense.json 2>> $O/r3c1_bench.err
MPCQP_DENSE=0 timeout 300 python bench.py --workload cfg2 > $O/r3c1_cfg2_sweeps.json 2>> $O/r3c1_bench.err
for B in 128 256 512; do
  timeout 300 python bench.py --batch $B --no-cpu-baseline --no-other-path --no-refactor-timing > $O/r3c1_bench_b$B.json 2>> $O/r3c1_bench.err
done
for B in 1 128 256 1024; do
  B=$B ITERS=100 MPCQP_LIB=scripts/diag/lib_timing.so timeout 300 python scripts/ablate.py >> $O/r3c1_timing.txt 2>&1
done
tail -5 $O/r3c1_tests.txt $O/r3c1_tests_nodense.txt; cat $O/r3c1_timing.txt | grep -v "^$" | tail -20
python - <<'PY'
import json
for f in ('r3c1_bench', 'r3c1_cfg2_dense', 'r3c1_cfg2_sweeps', 'r3c1_bench_b128', 'r3c1_bench_b256', 'r3c1_bench_b512'):
    try:
        d = json.load(open('gpurun_out/%s.json' % f))
        print(f, d.get('value'), d.get('unit'), 'ms/step', d.get('ms_per_step'), 'frac', (d.get('roofline') or {}).get('frac'), 'lat', json.dumps(d.get('latency'))[:600], 'cold', d.get('cold'))
    except Exception as e:
        print(f, 'ERR', e)
PY
