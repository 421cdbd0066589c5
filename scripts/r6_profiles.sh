#!/bin/bash
# round-6 evidence in one GPU call (from the repo root, on the GPU box): rocprofv3 kernel stats + FETCH / WRITE counters (a pass each) of the driver's
# bench command for cfg-3 at batch 1024 / 256 / 128 (the register-resident kernel: no L2-counter pass -- it hung once in round 5), of the same command
# with the bandwidth kernel forced (--backend sweeps, 1024 and 4096, with the L2 pass) and of cfg-5 with its leg's flags; SQ counters of the
# register-resident kernel at 128, 256 and 1024 instances.  Summaries: gpurun_out/r6*; scripts/r6_collect.py copies the ones to be judged into profiles/.
# usage: scripts/r6_profiles.sh [quick]     (quick: the headline command's kernel stats + FETCH / WRITE only)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
TCC=0 bash scripts/profile_counters.sh r6 cfg3 > $O/r6_profile_cfg3.log 2>&1
tail -n 4 $O/r6_profile_cfg3.log
[ "$1" == "quick" ] && exit 0
TCC=0 bash scripts/profile_counters.sh r6b256 cfg3 --batch 256 > $O/r6_profile_b256.log 2>&1
TCC=0 bash scripts/profile_counters.sh r6b128 cfg3 --batch 128 > $O/r6_profile_b128.log 2>&1
bash scripts/profile_counters.sh r6sw cfg3 --backend sweeps > $O/r6_profile_sw.log 2>&1
bash scripts/profile_counters.sh r6swb4096 cfg3 --backend sweeps --batch 4096 > $O/r6_profile_swb4096.log 2>&1
bash scripts/profile_counters.sh r6 cfg5 --steps 50 --warmup 25 > $O/r6_profile_cfg5.log 2>&1
bash scripts/pmc_sq.sh r6b128 128 > $O/r6_sq_b128.log 2>&1
bash scripts/pmc_sq.sh r6b256 256 > $O/r6_sq_b256.log 2>&1
bash scripts/pmc_sq.sh r6b1024 1024 > $O/r6_sq_b1024.log 2>&1
for f in b256 b128 sw swb4096 cfg5; do tail -n 3 $O/r6_profile_$f.log; done; tail -n 12 $O/r6_sq_b1024.log
( cd scripts/diag && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o mix_rate mix_rate.hip 2>/dev/null; timeout 60 ./mix_rate ) > $O/r6_mix_rate.txt 2>&1
cat $O/r6_mix_rate.txt
