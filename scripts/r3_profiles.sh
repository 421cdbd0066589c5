#!/bin/bash
# round-3 evidence: rocprofv3 kernel stats + HBM counters (separate passes) of the driver's bench command for cfg-3 and cfg-5, the same
# for the latency backend (256 instances), SQ counters of the plain-iteration harness for both backends, the final bench lines.
cd $GRAFT_REPO_ROOT; O=gpurun_out
bash scripts/profile_round.sh r3 cfg3 > $O/r3_profile_cfg3.log 2>&1
bash scripts/profile_round.sh r3 cfg5 > $O/r3_profile_cfg5.log 2>&1
bash scripts/profile_round.sh r3b256 cfg3 --batch 256 > $O/r3_profile_b256.log 2>&1
bash scripts/pmc_sq.sh r3_b1024 1024 > $O/r3_sq_b1024.txt 2>&1
bash scripts/pmc_sq.sh r3_b256 256 > $O/r3_sq_b256.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r3_bench_driver.json 2> $O/r3_bench_driver.err
timeout 900 python bench.py > $O/r3_bench_default.json 2> $O/r3_bench_default.err
timeout 600 python bench.py --workload cfg5 > $O/r3_bench_cfg5.json 2>> $O/r3_bench_default.err
timeout 600 python bench.py --workload cfg5 --batch 1024 --no-cpu-baseline > $O/r3_bench_cfg5_b1024.json 2>> $O/r3_bench_default.err
for B in 128 256 512; do timeout 300 python bench.py --batch $B --no-cpu-baseline > $O/r3_bench_b$B.json 2>> $O/r3_bench_default.err; done
timeout 300 python bench.py --workload cfg2 > $O/r3_bench_cfg2.json 2>> $O/r3_bench_default.err
tail -3 $O/r3_profile_cfg3.log $O/r3_profile_b256.log; tail -25 $O/r3_sq_b256.txt
