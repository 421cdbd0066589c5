"""Summarise rocprofv3 --pmc passes per kernel.

    python scripts/pmc_summary.py <bench.json of the FETCH pass> <f_counter_collection.csv> <w_counter_collection.csv> [<tcc csv>]

FETCH_SIZE and WRITE_SIZE are collected in SEPARATE runs (MI355X_MICROARCH.md, "rocprofv3 PMC slots") and are in KiB; on
gfx950 FETCH_SIZE counts 128-byte requests at 64 B, so wide coalesced reads are under-reported by 2x (same guide, HBM
section) -- both the raw and the doubled figure are printed.  The bench line printed by the profiled command carries the
ADMM iterations each k_mpc_run instantiation performed in the whole process (`accounting.process_totals`); counter totals
divided by them give the measured memory-side bytes per ADMM iteration per instance, which bench.py scales by its own
iteration count (`roofline.traffic`).  With PMC_JSON=<path> the per-kernel summary is also written as JSON."""
import collections
import csv
import json
import os
import sys

bench = json.load(open(sys.argv[1]))
totals = bench.get('accounting', {}).get('process_totals', {})
res = {}
for path in sys.argv[2:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row['Kernel_Name'].split('(')[0].replace('void ', '').replace(' ', '').strip()     # full template signature
            if not name.startswith(('k_', 'w8::k_')):          # (w8::: the 512-thread kernels of mpcqp_w8.hip)
                continue
            d = res.setdefault(name, collections.defaultdict(lambda: [0, 0.0, 0]))
            a = d[row['Counter_Name']]
            a[0] += 1; a[1] += float(row['Counter_Value']); a[2] += int(row['End_Timestamp']) - int(row['Start_Timestamp'])
summary = {}
for name, d in sorted(res.items()):
    nl = max(1, d['FETCH_SIZE'][0])
    f, w = d['FETCH_SIZE'][1] / nl, d['WRITE_SIZE'][1] / max(1, d['WRITE_SIZE'][0])
    s = {'launches': d['FETCH_SIZE'][0], 'fetch_kib_per_launch_raw': f, 'write_kib_per_launch_raw': w,
         'hbm_bytes_per_launch': (2 * f + w) * 1024, 'mean_duration_us_under_pmc': d['FETCH_SIZE'][2] / nl / 1e3,
         'note': 'FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B for wide coalesced reads); includes Infinity-Cache hits'}
    if name in totals and totals[name]['iters']:
        s['batch'] = bench.get('config', {}).get('batch_per_gpu')          # bench.py only applies the entry to runs of this batch
        s['admm_iters_all_launches'] = totals[name]['iters']
        s['hbm_bytes_per_iter_per_qp'] = s['hbm_bytes_per_launch'] * s['launches'] / totals[name]['iters']
    hit, miss = d.get('TCC_HIT_sum'), d.get('TCC_MISS_sum')
    if hit and miss and hit[1] + miss[1] > 0:
        s['l2_hit_rate'] = hit[1] / (hit[1] + miss[1])
    summary[name] = s
    print('%-40s launches %4d  FETCH %12.1f KiB/launch raw (x2: %9.1f MiB)  WRITE %10.1f KiB  %s%s' % (
        name[:40], s['launches'], f, 2 * f / 1024, w,
        ('%.0f B/iteration/QP  ' % s['hbm_bytes_per_iter_per_qp']) if 'hbm_bytes_per_iter_per_qp' in s else '',
        ('L2 hit rate %.3f' % s['l2_hit_rate']) if 'l2_hit_rate' in s else ''))
if os.environ.get('PMC_JSON'):
    json.dump(summary, open(os.environ['PMC_JSON'], 'w'), indent=1)
