#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wide.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/c12_wide.log
( MPCQP_LIB=pympc_amd/libmpcqp_timing.so timeout 300 python scripts/diag_wide.py 40 8 20 1 ) >> gpurun_out/c12_wide.log 2>&1
cat gpurun_out/c12_wide.log
