cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { python bench.py --workload cfg5 --steps 50 --warmup 25 --tuning $1 $2 --no-other-path --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('tuning', hex($1), '$2', 'value', int(d['value']), 'kernel_ms', r.get('kernel_ms'), 'frac', r.get('frac'))"; }
for rep in 1 2; do for q in 11 12 13; do run $((q<<24)); done; done
run $((12<<24)) "--steps 100 --warmup 50"
run $((13<<24)) "--steps 100 --warmup 50"
