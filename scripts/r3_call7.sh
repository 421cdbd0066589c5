#!/bin/bash
cd $GRAFT_REPO_ROOT
for B in 1 128; do
  B=$B ITERS=100 MPCQP_BCR=1 MPCQP_LIB=scripts/diag/lib_timing.so timeout 300 python scripts/ablate.py 2>&1 | grep -v amdgpu | tail -3
done
