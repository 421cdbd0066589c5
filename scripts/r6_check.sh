#!/bin/bash
# one GPU call that verifies a tree: the GPU suite, smoke(), the driver's bench command (its last stdout line is what the driver parses)
mkdir -p gpurun_out
TAG=${1:-r6}
timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/${TAG}_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.out 2> gpurun_out/${TAG}_bench_driver.err; echo "bench rc=$?" >> gpurun_out/${TAG}_bench_driver.err
tail -n 1 gpurun_out/${TAG}_bench_driver.out > gpurun_out/${TAG}_bench_driver.json
cp gpurun_out/bench_legs.json gpurun_out/${TAG}_bench_driver_legs.json 2>/dev/null
cat gpurun_out/${TAG}_gpu_tests.log; tail -3 gpurun_out/${TAG}_smoke.log; tail -2 gpurun_out/${TAG}_bench_driver.err
wc -c gpurun_out/${TAG}_bench_driver.json; cat gpurun_out/${TAG}_bench_driver.json
