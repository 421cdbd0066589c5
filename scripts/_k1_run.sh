cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_share_factor.py -q 2>&1 | grep -v "amdgpu.ids" | tail -15
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_sm.log 2> gpurun_out/bench_sm.err; tail -n 1 gpurun_out/bench_sm.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(len(json.dumps(d)), d['value'], d['roofline']['frac'], {k:v for k,v in d['legs'].items() if 'shared' in k})"
tail -3 gpurun_out/bench_sm.err
