#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
for d in 1 0; do for f in cart_pole point_mass; do
  MPCQP_DENSE=$d MPCQP_LIB=scripts/diag/lib_timing.so timeout 120 python scripts/diag_small.py $f 200 2>&1 | grep -v amdgpu.ids
done; done
timeout 300 python -m pytest tests/test_gpu_backends.py -m gpu -x -q 2>&1 | tail -3
