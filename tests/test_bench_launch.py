"""`python bench.py --gpus 2` must start two ranks by itself (no torch.distributed.run around it), prove that two ranks on
two distinct devices joined the process group, and go through scatter -> per-shard work -> all-gather.  Here on CPU: the
same entry with the collective backend overridden to gloo and the solver left out (--dry-run); on the GPU box the driver
runs the real thing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *flags):
    env = dict(os.environ, MPCQP_BENCH_BACKEND='gloo', **extra_env)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dry-run'] + list(flags), env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout          # rank 0 prints ONE JSON line
    assert r.stdout.rstrip().splitlines()[-1] == lines[0], r.stdout[-1500:]      # ... and it is the LAST line
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_bench_gpus2_self_launches_two_ranks():
    out = _run({}, '--gpus', '2')
    assert out['n_gpus'] == 2 and out['ranks_seen'] == 2 and out['devices_seen'] == 2
    assert out['backend'] == 'gloo' and out['gathered_ok'] is True


@pytest.mark.timeout(300)
def test_bench_gpus1_stays_in_process():
    out = _run({}, '--gpus', '1')
    assert out['n_gpus'] == 1 and out['ranks_seen'] == 1 and out['devices_seen'] == 1 and out['gathered_ok'] is True


@pytest.mark.timeout(300)
def test_forced_process_group_of_one_rank_runs_the_collectives():
    """MPCQP_BENCH_FORCE_PG=1: a single rank still builds the process group and goes through scatter / all-gather / barrier (here gloo;
    with RCCL on the GPU box: tests/test_gpu_rccl_single_rank.py)."""
    out = _run({'MPCQP_BENCH_FORCE_PG': '1'}, '--gpus', '1')
    assert out['n_gpus'] == 1 and out['ranks_seen'] == 1 and out['backend'] == 'gloo' and out['gathered_ok'] is True


@pytest.mark.timeout(300)
def test_bench_gpus4_total_batch_1024_is_the_strong_scaling_command():
    """`bench.py --gpus 4 --total-batch 1024` (BASELINE configs[3] read literally, on four ranks): 256 instances each, one packed scatter, the
    all-gathers, a per-rank report in the line."""
    out = _run({}, '--gpus', '4', '--total-batch', '1024')
    assert out['n_gpus'] == 4 and out['ranks_seen'] == 4 and out['devices_seen'] == 4 and out['total_batch'] == 1024 and out['gathered_ok'] is True
    assert [(r['rank'], r['instances'], r['first_instance']) for r in out['per_rank']] == [(i, 256, 256 * i) for i in range(4)]


@pytest.mark.timeout(300)
def test_bench_total_batch_need_not_divide():
    out = _run({}, '--gpus', '4', '--total-batch', '10')
    assert out['gathered_ok'] is True and [r['instances'] for r in out['per_rank']] == [3, 3, 3, 1]


@pytest.mark.timeout(300)
def test_bench_shared_model_broadcasts_one_model_and_scatters_the_states():
    """`bench.py --gpus 3 --shared-model --total-batch 8`: ONE model broadcast from rank 0 (sharding.broadcast_model), the states alone scattered
    (last paragraph), a short last rank."""
    out = _run({}, '--gpus', '3', '--shared-model', '--total-batch', '8')
    assert out['n_gpus'] == 3 and out['ranks_seen'] == 3 and out['shared_model'] is True and out['gathered_ok'] is True
    assert [r['instances'] for r in out['per_rank']] == [3, 3, 2]
