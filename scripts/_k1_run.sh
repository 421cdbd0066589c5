cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { python bench.py --workload $3 --steps $4 --warmup $5 --tuning $1 $2 --no-other-path --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$3 tuning', hex($1), '$2', 'value', int(d['value']), 'kernel_ms', r.get('kernel_ms'), 'frac', r.get('frac'), 'iters', d.get('mean_admm_iters'))"; }
run 0 "" cfg5 50 25
run 0 "" cfg5 50 25
run 0 "--batch 700" cfg5 50 25
for q in 0 12 14; do run $((q<<24)) "--backend sweeps" cfg3 20 5; done
for q in 0 12 14; do run $((q<<24)) "--backend sweeps" cfg3 100 20; done
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
