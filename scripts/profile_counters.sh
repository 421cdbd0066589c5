#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_counters.sh <tag> <workload> [extra bench flags, e.g. --batch 256 | --steps 50 --warmup 25]
# Device-loop path of the driver's bench command: kernel stats, the two byte counters (FETCH_SIZE, WRITE_SIZE) and the L2 hit / miss counters,
# each in a run of its own under `timeout` (MI355X_MICROARCH.md: counters in their own passes, --kernel-trace only).
# Summary: gpurun_out/<tag>_<workload>_device_loop_pmc_hbm_traffic.txt and _pmc.json.   TCC=0 skips the L2 pass.
tag=${1:-r5}; wl=${2:-cfg3}; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
T=${tag}_${wl}_device_loop
STEPS="--steps 20 --warmup 5"
case "$*" in *--steps*) STEPS="";; esac
CMD="python $R/bench.py $STEPS --no-cpu-baseline --no-other-path --no-refactor-timing --workload $wl --path device_loop $*"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -o ks -- $CMD > $O/${T}_bench_under_rocprof.json 2> $O/${T}_stats.log
cp $O/${T}_stats/ks_kernel_stats.csv $O/${T}_kernel_stats.csv
# (the full record of the FETCH pass -- bench.py's side file -- carries the per-kernel iteration totals pmc_summary.py divides by; stdout is the compact line)
MPCQP_BENCH_LEGS=$O/${T}_bench_under_pmc.json timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${T}_pmc_f -o f -- $CMD > /dev/null 2> $O/${T}_pmc_f.log
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${T}_pmc_w -o w -- $CMD > /dev/null 2> $O/${T}_pmc_w.log
TCCCSV=""
if [ "${TCC:-1}" != "0" ]; then
  timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/${T}_pmc_t -o t -- $CMD > /dev/null 2> $O/${T}_pmc_t.log && TCCCSV=$O/${T}_pmc_t/t_counter_collection.csv
fi
PMC_JSON=$O/${T}_pmc.json python $R/scripts/pmc_summary.py $O/${T}_bench_under_pmc.json $O/${T}_pmc_f/f_counter_collection.csv $O/${T}_pmc_w/w_counter_collection.csv $TCCCSV > $O/${T}_pmc_hbm_traffic.txt
head -3 $O/${T}_kernel_stats.csv | cut -c1-200; cat $O/${T}_pmc_hbm_traffic.txt
