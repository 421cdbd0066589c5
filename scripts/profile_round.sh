#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_round.sh <tag> [workload] [extra bench flags]
# (give runs with extra flags their own tag, e.g. `profile_round.sh r2b4096 cfg3 --batch 4096`: the output names carry tag and workload only)
# For both bench paths of <workload> (cfg3 | cfg5): rocprofv3 kernel stats + the PMC passes (FETCH_SIZE, WRITE_SIZE and
# TCC_HIT_sum/TCC_MISS_sum each in a run of its own, with --kernel-trace only, as MI355X_MICROARCH.md prescribes) of the
# bench command the driver uses.  Every rocprofv3 run sits under `timeout`: in round 4 one TCC pass (batch 4096, stepwise) hung and took
# forty GPU-minutes with it.  Summaries land in gpurun_out/<tag>_<workload>_<path>_*; copy the ones to be judged into
# profiles/ (profiles/pmc_hbm_traffic.json = gpurun_out/<tag>_pmc_hbm_traffic.json merged per workload).
tag=${1:-r2}; wl=${2:-cfg3}; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for path in device_loop stepwise; do
  T=${tag}_${wl}_${path}
  CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-path --no-refactor-timing --workload $wl --path $path $*"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -o ks -- $CMD > $O/${T}_bench_under_rocprof.json 2> $O/${T}_stats.log
  cp $O/${T}_stats/ks_kernel_stats.csv $O/${T}_kernel_stats.csv
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${T}_pmc_f -o f -- $CMD > $O/${T}_bench_under_pmc.json 2> $O/${T}_pmc_f.log
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${T}_pmc_w -o w -- $CMD > /dev/null 2> $O/${T}_pmc_w.log
  timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/${T}_pmc_t -o t -- $CMD > /dev/null 2> $O/${T}_pmc_t.log
  PMC_JSON=$O/${T}_pmc.json python $R/scripts/pmc_summary.py $O/${T}_bench_under_pmc.json $O/${T}_pmc_f/f_counter_collection.csv \
      $O/${T}_pmc_w/w_counter_collection.csv $O/${T}_pmc_t/t_counter_collection.csv > $O/${T}_pmc_hbm_traffic.txt
  head -4 $O/${T}_kernel_stats.csv | cut -c1-160; cat $O/${T}_pmc_hbm_traffic.txt
done
python - <<PY
import json, os
f = '$O/${tag}_pmc_hbm_traffic.json'
out = json.load(open(f)) if os.path.exists(f) else {}
out['$wl'] = {p: json.load(open('$O/${tag}_${wl}_%s_pmc.json' % p)) for p in ('device_loop', 'stepwise')}
json.dump(out, open(f, 'w'), indent=1)
PY
