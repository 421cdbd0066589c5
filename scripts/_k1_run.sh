cd $GRAFT_REPO_ROOT
run() { python bench.py --path stepwise --steps 40 --warmup 20 --tuning $1 --no-other-path --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('stepwise tuning', hex($1), 'value', int(d['value']), 'ms/step', d['ms_per_step'], 'kernel_ms', r.get('kernel_ms'), 'launches', r.get('launches'), 'iters', d.get('mean_admm_iters'))"; }
run 0; run $(((1<<29)|16)); run 0; run $(((1<<29)|16))
