#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_round.sh <tag>
# For both bench paths: rocprofv3 kernel stats + the two PMC passes (separate runs, as MI355X_MICROARCH.md
# prescribes) of the default bench command, then the un-profiled bench line.  Summaries land in
# gpurun_out/<tag>_<path>_*; copy the ones to be judged into profiles/.
tag=${1:-r1}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for path in device_loop stepwise; do
  T=${tag}_${path}
  CMD="python $R/bench.py --no-cpu-baseline --no-other-path --path $path"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -o ks -- $CMD > $O/${T}_bench_under_rocprof.json 2> $O/${T}_stats.log
  cp $O/${T}_stats/ks_kernel_stats.csv $O/${T}_kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${T}_pmc_f -o f -- $CMD > /dev/null 2> $O/${T}_pmc_f.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${T}_pmc_w -o w -- $CMD > /dev/null 2> $O/${T}_pmc_w.log
  PMC_JSON=$O/${T}_pmc.json python $R/scripts/pmc_summary.py $O/${T}_pmc_f/f_counter_collection.csv $O/${T}_pmc_w/w_counter_collection.csv > $O/${T}_pmc_hbm_traffic.txt
  head -3 $O/${T}_kernel_stats.csv | cut -c1-160; head -2 $O/${T}_pmc_hbm_traffic.txt
done
python - <<PY
import json
out = {p: json.load(open('$O/${tag}_%s_pmc.json' % p)) for p in ('device_loop', 'stepwise')}
json.dump(out, open('$O/${tag}_pmc_hbm_traffic.json', 'w'), indent=1)
PY
python $R/bench.py > $O/${tag}_bench.json 2> /dev/null
cut -c1-200 $O/${tag}_bench.json
