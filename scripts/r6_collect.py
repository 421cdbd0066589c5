"""After scripts/r6_profiles.sh: copy the summaries to be judged from gpurun_out/ into profiles/ and merge the counter summaries into
profiles/pmc_hbm_traffic.json (one key per profiled command shape; tests/test_roofline_model.py and bench.py's roofline.traffic read it)."""
import json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, 'gpurun_out'), os.path.join(R, 'profiles')
shapes = {'cfg3': 'r6_cfg3', 'cfg3_sweeps': 'r6sw_cfg3', 'cfg3_sweeps_b4096': 'r6swb4096_cfg3', 'cfg3_b256': 'r6b256_cfg3', 'cfg3_b128': 'r6b128_cfg3',
          'cfg5': 'r6_cfg5'}
old = json.load(open(os.path.join(P, 'pmc_hbm_traffic.json')))
new = {'round': '6: every device_loop entry taken on the final tree of round 6 with scripts/r6_profiles.sh (FETCH_SIZE and WRITE_SIZE in a pass each; TCC_HIT/MISS as well for '
                'the bandwidth kernels and cfg-5; cfg5 with its bench leg\'s flags, 25 warm-up + 50 timed steps); stepwise entries are the round-4 profile'}
for key, tag in shapes.items():
    t = tag + '_device_loop'
    pj = os.path.join(O, t + '_pmc.json')
    if not os.path.exists(pj):
        print('missing', pj); new[key] = old.get(key); continue
    new[key] = {'device_loop': json.load(open(pj))}
    if key in old and 'stepwise' in old[key]:
        new[key]['stepwise'] = old[key]['stepwise']
    for suffix in ('_kernel_stats.csv', '_pmc_hbm_traffic.txt', '_bench_under_rocprof.json'):
        src = os.path.join(O, t + suffix)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(P, t + suffix))
json.dump(new, open(os.path.join(P, 'pmc_hbm_traffic.json'), 'w'), indent=1)
for f in ('r6b128_sq_counters.txt', 'r6b256_sq_counters.txt', 'r6b1024_sq_counters.txt', 'r6_mix_rate.txt', 'r6_makespan.txt'):
    if os.path.exists(os.path.join(O, f)):
        shutil.copy(os.path.join(O, f), os.path.join(P, f))
for key in shapes:
    e = (new.get(key) or {}).get('device_loop') or {}
    for k, v in e.items():
        if 'k_mpc_run' in k and k.replace(' ', '').endswith('true>') and 'hbm_bytes_per_iter_per_qp' in v:
            print('%-12s %-44s batch %5s  %9.0f B / iteration / QP   L2 hit %s' % (key, k, v.get('batch'), v['hbm_bytes_per_iter_per_qp'], v.get('l2_hit_rate')))
