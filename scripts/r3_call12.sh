#!/bin/bash
mkdir -p gpurun_out
( MPCQP_LIB=pympc_amd/libmpcqp_timing.so timeout 300 python scripts/diag_factor.py ) > gpurun_out/c12_factor.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 >> gpurun_out/c12_factor.log
cat gpurun_out/c12_factor.log
