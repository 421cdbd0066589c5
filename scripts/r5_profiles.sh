#!/bin/bash
# round-5 evidence in one GPU call (from the repo root, on the GPU box): rocprofv3 kernel stats + FETCH / WRITE / L2 counters (a pass each) of the
# driver's bench command for cfg-3 (batch 1024, 4096, 256, 128; 1024 and 4096 also with the bandwidth kernel forced: --backend sweeps) and of cfg-5 with its LEG's flags (25 warm-up + 50 timed steps: the warm 25-step
# launches); SQ counters of the latency kernel at 128 and 256 instances; the whole-chip read-only stream rates.  Summaries: gpurun_out/r5*; the
# ones to be judged are copied into profiles/ (scripts/r5_collect.py).
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
( cd scripts/diag && timeout 120 ./hbm_stream ) > $O/r5_hbm_stream.txt 2>&1
bash scripts/profile_counters.sh r5 cfg3 > $O/r5_profile_cfg3.log 2>&1
# (the library's choice at 4096 instances -- the register-resident kernel -- is not profiled: its L2-counter pass hung on 2026-09-27 and took the round's GPU budget with it)
bash scripts/profile_counters.sh r5sw cfg3 --backend sweeps > $O/r5_profile_sw.log 2>&1
bash scripts/profile_counters.sh r5swb4096 cfg3 --backend sweeps --batch 4096 > $O/r5_profile_swb4096.log 2>&1
bash scripts/profile_counters.sh r5b256 cfg3 --batch 256 > $O/r5_profile_b256.log 2>&1
bash scripts/profile_counters.sh r5b128 cfg3 --batch 128 > $O/r5_profile_b128.log 2>&1
bash scripts/profile_counters.sh r5 cfg5 --steps 50 --warmup 25 > $O/r5_profile_cfg5.log 2>&1
bash scripts/pmc_sq.sh r5b128 128 > $O/r5_sq_b128.log 2>&1
bash scripts/pmc_sq.sh r5b256 256 > $O/r5_sq_b256.log 2>&1
for f in cfg3 sw swb4096 b256 b128 cfg5; do tail -n 3 $O/r5_profile_$f.log; done; tail -n 12 $O/r5_sq_b128.log; cat $O/r5_hbm_stream.txt
