#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_round.sh <tag>
# rocprofv3 kernel stats + the two PMC passes (separate runs, as MI355X_MICROARCH.md prescribes) of the default
# bench command; summaries land in gpurun_out/<tag>_* -- copy the ones to be judged into profiles/.
tag=${1:-r1}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --no-cpu-baseline --no-other-path"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o ks -- $CMD > $O/${tag}_bench_under_rocprof.json 2> $O/${tag}_stats.log
cp $O/${tag}_stats/ks_kernel_stats.csv $O/${tag}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${tag}_pmc_f -o f -- $CMD > /dev/null 2> $O/${tag}_pmc_f.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${tag}_pmc_w -o w -- $CMD > /dev/null 2> $O/${tag}_pmc_w.log
PMC_JSON=$O/${tag}_pmc_hbm_traffic.json python $R/scripts/pmc_summary.py $O/${tag}_pmc_f/f_counter_collection.csv $O/${tag}_pmc_w/w_counter_collection.csv > $O/${tag}_pmc_hbm_traffic.txt
$CMD > $O/${tag}_bench.json 2>/dev/null
head -5 $O/${tag}_kernel_stats.csv | cut -c1-160; cat $O/${tag}_pmc_hbm_traffic.txt | head -4; cut -c1-200 $O/${tag}_bench.json
