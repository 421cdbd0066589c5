"""The ONE line bench.py hands the driver: compact (a few kB), numbers and short identifiers only, JSON round trip, every key of the bench
contract present.  Round 5's line was 27.5 kB (every leg twice, with its prose) and the driver recorded `parsed: null`; this holds the
assembled line to bench.LINE_BUDGET on canned records -- the round-5 record of the driver's own command, and a synthetic 8-rank one."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
            'config', 'roofline', 'cpu_baseline')
ROOF = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')


def canned():
    """profiles/r5_bench_driver.json (what the driver's command printed in round 5: every leg present) re-keyed the way this round's roofline()
    names things."""
    with open(os.path.join(ROOT, 'profiles', 'r5_bench_driver.json')) as f:
        out = json.load(f)
    ro = out['roofline']
    ro.pop('legs', None)
    ro.update(bound='hbm', unit='GB/s', peak=8000.0, achieved=950.1234567, frac=0.11876543, frac_counters=0.149, traffic_GBps=1190.0,
              mfma_issue_frac=0.2031, useful_flop_frac=0.0508, bytes_per_launch=1.2e10, model_bytes_per_iter_per_qp=15500.0,
              traffic_over_model=1.26, traffic_profile='profiles/pmc_hbm_traffic.json:cfg3/device_loop (batch 1024)')
    return out


def check_line(line, n_gpus):
    txt = json.dumps(line)
    assert len(txt) < bench.LINE_BUDGET, len(txt)
    back = json.loads(txt)
    for k in CONTRACT:
        assert k in back, k
    for k in ROOF:
        assert k in back['roofline'], k
    assert back['roofline']['bound'] in ('hbm', 'mfma')
    assert abs(back['roofline']['frac'] - back['roofline']['achieved'] / back['roofline']['peak']) < 1e-3
    assert set(('value', 'unit', 'cores', 'kind', 'sample')) <= set(back['cpu_baseline'])
    assert back['n_gpus'] == n_gpus and len(back['per_rank']) == n_gpus
    assert 'workload' in back['config'] and 'model' not in back['config']
    # no prose: nothing longer than a kernel name or the workload / sample strings
    def walk(o, path=''):
        if isinstance(o, dict):
            for k, v in o.items():
                walk(v, path + '/' + k)
        elif isinstance(o, list):
            for i, v in enumerate(o):
                walk(v, path + '/%d' % i)
        elif isinstance(o, str):
            limit = 170 if path.endswith(('/workload', '/sample', '/parallelism', '/metric')) else 80
            assert len(o) <= limit, (path, len(o))
    walk(back)
    return back


def test_round5_record_fits():
    out = canned()
    out['legs_file'] = 'gpurun_out/bench_legs.json'
    assert len(json.dumps(out)) > 20000                     # (the full record is what used to be printed)
    back = check_line(bench.compact_line(out), 1)
    assert back['value'] == float('%.6g' % out['value'])
    legs = back['legs']
    for k in ('stepwise', 'eps_1e-9', 'cfg5_b512', 'b128', 'b256', 'bandwidth_kernel_b1024', 'sweeps_b4096', 'latency_cfg2', 'latency_notebook', 'projection_8gpu'):
        assert k in legs, (k, sorted(legs))
    assert legs['b128']['value'] > 1e5 and legs['cfg5_b512']['value'] > 1e4
    assert back['u_err']['eps_1e-09']['max_rel'] < 1e-6
    assert back['cpu_baseline']['all_cores']['cores'] >= 1


def test_eight_rank_record_fits():
    out = canned()
    r0 = out['per_rank'][0]
    out['per_rank'] = [dict(r0, rank=r, first_instance=1024 * r, roofline_frac=0.118765432, scatter_ms=1.234567, gather_calls=1, gather_ms_per_call=0.0456789) for r in range(8)]
    out.update(n_gpus=8, ranks_seen=8, devices_seen=8, collective_backend='nccl',
               strong_scaling={'scaling': 'strong', 'total_batch': 1024, 'batch_per_gpu': 128, 'path': 'device_loop', 'value': 5.8e6, 'ms_per_step': 0.1765, 'mean_admm_iters': 37.9})
    back = check_line(bench.compact_line(out), 8)
    assert back['ranks_seen'] == 8 and back['devices_seen'] == 8 and back['collective_backend'] == 'nccl'
    assert [r['rank'] for r in back['per_rank']] == list(range(8))
    assert all('value' in r and 'roofline_frac' in r and 'scatter_ms' in r for r in back['per_rank'])
    assert back['legs']['strong_total1024']['value'] == 5.8e6


def test_budget_sheds_detail_not_contract():
    out = canned()
    out['small_batch_legs'] = {'leg%03d' % i: dict(out['small_batch_legs']['b128']) for i in range(200)}      # far too many legs
    line = bench.compact_line(out)
    assert len(json.dumps(line)) < bench.LINE_BUDGET
    for k in CONTRACT:
        assert k in line, k


def test_emit_writes_side_file_and_prints_last(tmp_path, monkeypatch, capsys):
    out = canned()
    monkeypatch.setenv('MPCQP_BENCH_LEGS', str(tmp_path / 'legs.json'))
    bench.emit(out)
    printed = capsys.readouterr().out.strip().splitlines()
    last = json.loads(printed[-1])
    assert last['metric'] == out['metric'] and len(printed[-1]) < bench.LINE_BUDGET
    with open(tmp_path / 'legs.json') as f:
        full = json.load(f)
    assert full['hbm_leg']['roofline']['kernel'] == out['hbm_leg']['roofline']['kernel']      # everything is in the side file
