#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_final.sh <tag>
# rocprofv3 kernel stats (no counters) of the driver's bench command on the device-loop path, cfg-3 and cfg-5, each run under `timeout`:
# the per-kernel durations of the tree as it stands, to set beside the bench line's HIP-event figure.  (The counter passes live in profile_round.sh.)
tag=${1:-r4f}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for wl in cfg3 cfg5; do
  T=${tag}_${wl}_device_loop
  CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-path --no-refactor-timing --workload $wl --path device_loop"
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -o ks -- $CMD > $O/${T}_bench_under_rocprof.json 2> $O/${T}_stats.log
  cp $O/${T}_stats/ks_kernel_stats.csv $O/${T}_kernel_stats.csv
  head -4 $O/${T}_kernel_stats.csv | cut -c1-200
done
