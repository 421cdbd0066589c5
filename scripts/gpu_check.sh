#!/bin/bash
# full verification of the tree: the GPU suite, smoke(), the default bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/check_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/check_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/check_smoke.log
timeout 900 python bench.py > gpurun_out/check_bench_default.json 2> gpurun_out/check_bench_default.err; echo "bench rc=$?" >> gpurun_out/check_bench_default.err
cat gpurun_out/check_gpu_tests.log; tail -3 gpurun_out/check_smoke.log; tail -2 gpurun_out/check_bench_default.err; cut -c1-600 gpurun_out/check_bench_default.json
