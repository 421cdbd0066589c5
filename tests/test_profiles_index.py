"""The evidence under profiles/ that bench.py and the docs point at exists: every counter-profile key bench.py looks up (roofline.traffic of the headline, of the
forced bandwidth-kernel legs, of the small-batch legs and of cfg-5) has an entry for a closed-loop k_mpc_run kernel with bytes per iteration and QP, and every
round-6 file the index (profiles/README.md) names is there."""
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')


def test_counter_profile_has_the_keys_bench_looks_up():
    prof = json.load(open(os.path.join(P, 'pmc_hbm_traffic.json')))
    want = {'cfg3': ('w8::k_mpc_run<16,true,12,4,231,true>', 1024), 'cfg3_sweeps': ('k_mpc_run<16,true,12,4,0,true>', 1024),
            'cfg3_sweeps_b4096': ('k_mpc_run<16,true,12,4,0,true>', 4096), 'cfg3_b256': ('w8::k_mpc_run<16,true,12,4,231,true>', 256),
            'cfg3_b128': ('w8::k_mpc_run<16,true,12,4,231,true>', 128), 'cfg5': ('k_mpc_run<32,false,20,8,0,true>', 512)}
    for key, (kernel, batch) in want.items():
        entry = prof[key]['device_loop'][kernel]
        assert entry['batch'] == batch and entry['hbm_bytes_per_iter_per_qp'] > 0, (key, entry)


def test_files_named_in_the_index_exist():
    text = open(os.path.join(P, 'README.md')).read()
    current = text.split('`r1/`, `r2_*`')[0]
    names = set(re.findall(r'`(r6[a-z0-9]*_[A-Za-z0-9_.*]+)`', current))
    assert names, 'no round-6 entries found in profiles/README.md'
    files = os.listdir(P)
    for n in sorted(names):
        if n.endswith('*'):
            assert any(f.startswith(n[:-1]) for f in files), n
        elif '.' in n:
            assert n in files, n


def test_bench_lines_of_the_round_are_the_final_kernels():
    """The driver's command on the final tree: the compact line (what the driver parses) and the full record beside it."""
    d = json.load(open(os.path.join(P, 'r6_bench_driver.json')))
    assert d['steps'] == 20 and d['warmup'] == 5 and d['n_gpus'] == 1
    assert len(json.dumps(d)) < 6000
    ro = d['roofline']
    assert ro['kernel'] == 'w8::k_mpc_run<16,true,12,4,231,true>' and ro['bound'] == 'hbm' and 0.0 < ro['frac'] < 1.0
    assert abs(ro['frac'] - ro['achieved'] / ro['peak']) < 1e-3 and ro['traffic'] > 0
    legs = d['legs']
    for k in ('bandwidth_kernel_b1024', 'sweeps_b4096', 'cfg5_b512', 'cfg5_b1024', 'b128', 'b256', 'latency_cfg2', 'latency_notebook', 'latency_kalman_np200', 'stepwise', 'eps_1e-9'):
        assert k in legs, k
    assert d['cpu_baseline']['kind'] == 'port' and d['cpu_baseline']['cores'] == 1
    full = json.load(open(os.path.join(P, 'r6_bench_driver_legs.json')))
    assert full['value'] == pytest.approx(d['value'], rel=1e-5) and full['roofline']['kernel'] == ro['kernel']
    assert full['cfg5_leg']['roofline']['bound'] == 'hbm' and full['small_batch_legs']['bandwidth_kernel_b1024']['roofline']['bound'] == 'hbm'
