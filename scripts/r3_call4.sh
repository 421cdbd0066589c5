#!/bin/bash
cd $GRAFT_REPO_ROOT
MPCQP_LIB=scripts/diag/lib_timing.so timeout 120 python scripts/diag_small.py cart_pole 200 2>&1 | grep -v amdgpu.ids | tail -3
