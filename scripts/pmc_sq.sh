#!/bin/bash
# usage (on the GPU box, from the repo root): [EXTRA="--backend sweeps --shared-model"] scripts/pmc_sq.sh <tag> <batch> [workload]      (EXTRA: more bench flags)
# SQ counters of the solve kernel under the driver's bench command at that batch (device loop), four passes of five counters each:
# where the waves' cycles go -- issuing (ACTIVE_INST_*), parked at s_waitcnt / barriers (WAIT_ANY), stalled at issue (WAIT_INST_ANY), matrix pipe busy
# (VALU_MFMA_BUSY_CYCLES), LDS bank conflicts.  Summary: gpurun_out/<tag>_sq_counters.txt (mean per launch of k_mpc_run).
tag=$1; B=${2:-128}; wl=${3:-cfg3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-path --no-refactor-timing --workload $wl --path device_loop --batch $B ${EXTRA:-}"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/sq_${tag}_$i -o c -- $CMD > $O/sq_${tag}_$i.log 2>&1
done
python - > $O/${tag}_sq_counters.txt <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
names = set()
for f in sorted(glob.glob('$O/sq_${tag}_*/c_counter_collection.csv')):
    for row in csv.DictReader(open(f)):
        if 'k_mpc_run' not in row['Kernel_Name'] or 'true>' not in row['Kernel_Name'].replace(' ', ''): continue      # the closed-loop instantiation
        names.add(row['Kernel_Name'].split('(')[0].replace('void ', '').replace(' ', ''))
        a = agg[row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
print('# SQ counters, %s, batch $B, workload $wl: mean per launch (warm-up and timed launch) of %s' % ('$tag', ', '.join(sorted(names))))
m = {k: t / n for k, (n, t) in agg.items()}
for k in sorted(m): print('%-28s launches %3d  mean %16.1f' % (k, agg[k][0], m[k]))
wc = m.get('SQ_WAVE_CYCLES')
if wc:
    for k in ('SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_LDS_BANK_CONFLICT', 'SQ_INST_CYCLES_VMEM'):
        if k in m: print('%-28s / SQ_WAVE_CYCLES = %.3f' % (k, m[k] / wc))
PY
cat $O/${tag}_sq_counters.txt
