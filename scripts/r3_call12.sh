#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_seam.py -q -x 2>&1 | tail -25 > gpurun_out/c12_seam.log
cat gpurun_out/c12_seam.log
