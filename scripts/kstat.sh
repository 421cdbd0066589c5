#!/bin/bash
# usage: scripts/kstat.sh <tag> [env...]  -- rocprofv3 kernel stats of a short bench run (development tool)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ks_$tag -o ks -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/ks_$tag.log 2>&1
grep -o '"value": [0-9.]*' $GRAFT_REPO_ROOT/gpurun_out/ks_$tag.log | head -1
head -4 $GRAFT_REPO_ROOT/gpurun_out/ks_$tag/ks_kernel_stats.csv | tail -3 | cut -c1-120
