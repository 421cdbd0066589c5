#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wide.py -m gpu -q -x 2>&1 | tail -40 > gpurun_out/c11_wide.log
cat gpurun_out/c11_wide.log
