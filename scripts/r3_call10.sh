#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_backends.py -m gpu -x -q 2>&1 | tail -2
for B in 128 256 512; do timeout 300 python bench.py --batch $B --no-cpu-baseline --no-other-path --no-refactor-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B', d['config']['batch_per_gpu'], round(d['value']), round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['frac'],4))"; done
timeout 300 python bench.py --workload cfg2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read())['latency']; print({k: (round(v,1) if isinstance(v,float) else v) for k,v in d.items()})"
