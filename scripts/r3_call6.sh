#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_backends.py -m gpu -x -q 2>&1 | tail -15
for B in 128 256; do
  timeout 300 python bench.py --batch $B --no-cpu-baseline --no-other-path --no-refactor-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B', d['config']['batch_per_gpu'], d['value'], d['ms_per_step'], d['mean_admm_iters'], d['roofline']['kernel'], d['cold'])"
done
for B in 128; do
  B=$B ITERS=100 MPCQP_LIB=scripts/diag/lib_timing.so timeout 300 python scripts/ablate.py 2>&1 | grep -v amdgpu | tail -3
done
