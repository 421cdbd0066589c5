#!/bin/bash
# usage: scripts/pmc_sq.sh <tag> <B>  -- SQ counters of the plain-iteration harness (development tool)
tag=$1; B=${2:-1024}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  B=$B ITERS=100 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/sq_${tag}_$i -o c -- python $R/scripts/ablate.py > $O/sq_${tag}_$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('$O/sq_${tag}_*/c_counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        if 'k_mpc_run' not in row['Kernel_Name']: continue
        a = agg[row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
for k, (n, t) in sorted(agg.items()): print('%-28s launches %3d  mean %16.1f' % (k, n, t / n))
PY
