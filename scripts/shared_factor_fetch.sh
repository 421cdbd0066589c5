#!/bin/bash
# on the GPU box, from the repo root: FETCH_SIZE of the closed-loop launches of scripts/shared_factor_rate.py (own factors, then ONE shared factor, then the
# register-resident kernel) at 4096 instances -- a pass of its own, --kernel-trace only (MI355X_MICROARCH.md).  Summary: gpurun_out/r6_shared_factor_fetch.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/shf_pmc -o f -- python $R/scripts/shared_factor_rate.py --batch ${1:-4096} --backend sweeps > $O/shf_run.log 2> $O/shf_pmc.log
python - $O/shf_pmc/f_counter_collection.csv <<'P' > $O/r6_shared_factor_fetch.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r['Counter_Name'] == 'FETCH_SIZE' and 'k_mpc_run' in r['Kernel_Name'] and 'true>(' in r['Kernel_Name'].replace(' ', '')]
rows.sort(key=lambda r: int(r['Dispatch_Id']))
print('closed-loop launches (20 steps each) of scripts/shared_factor_rate.py in dispatch order: per run one warm-up and two timed launches; runs: own factors, ONE shared factor, register-resident kernel')
print('FETCH_SIZE in KiB as counted (gfx950 counts wide coalesced reads at half their size: x2 for bytes; includes Infinity-Cache hits)')
for r in rows:
    v = float(r['Counter_Value'])
    print('dispatch %6s  %-48s FETCH %12.0f KiB raw  (x2: %8.1f MiB)' % (r['Dispatch_Id'], r['Kernel_Name'].replace('void ', '').replace(' ', '').replace('(RunKArgs)', '')[:48], v, 2 * v / 1024))
P
grep -v amdgpu.ids $O/shf_run.log | cut -c1-400; cat $O/r6_shared_factor_fetch.txt
