"""The RCCL branch of bench.py and pympc_amd/sharding.py on a box with ONE GPU: MPCQP_BENCH_FORCE_PG=1 makes a single rank build the
`nccl` (= RCCL) process group on its device and run what N > 1 ranks run around the timed region -- scatter of the problem data, barrier,
all-gather of u* (per step on the stepwise path, per launch on the device loop), max-reduction of the elapsed time -- with a communicator
of one.  Not a scaling measurement (the driver's 8-GPU run is that); it shows the branch executes on the hardware and leaves the results alone."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ['--gpus', '1', '--steps', '4', '--warmup', '2', '--batch', '64', '--no-cpu-baseline', '--no-other-path', '--no-refactor-timing']


def _bench(force, *flags):
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'MPCQP_BENCH_BACKEND', 'MPCQP_BENCH_FORCE_PG'):
        env.pop(k, None)
    if force:
        env['MPCQP_BENCH_FORCE_PG'] = '1'
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + FLAGS + list(flags), env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    # the driver parses the LAST stdout line: RCCL's banner (C stdio, written when its buffer is flushed) must not land behind it
    assert r.stdout.rstrip().splitlines()[-1] == lines[0], r.stdout[-1500:]
    return json.loads(lines[0])


@pytest.mark.timeout(600)
@pytest.mark.parametrize('path', ['device_loop', 'stepwise'])
def test_single_rank_rccl_group_runs_and_changes_nothing(path):
    forced, plain = _bench(True, '--path', path), _bench(False, '--path', path)
    assert forced['collective_backend'] == 'nccl' and forced['ranks_seen'] == 1 and forced['devices_seen'] == 1 and forced['n_gpus'] == 1
    assert plain.get('collective_backend') is None
    assert forced['value'] > 0 and forced['steps'] == 4
    # the same seeded workload with and without the process group: the same ADMM work, iteration for iteration
    assert forced['mean_admm_iters'] == plain['mean_admm_iters']


@pytest.mark.timeout(600)
def test_shared_model_broadcast_over_rccl_with_a_communicator_of_one():
    """bench.py --shared-model: ONE controller broadcast (sharding.broadcast_model: a packed RCCL broadcast), the states alone scattered, every instance of the
    shard on one shared KKT factor -- and the same ADMM work with and without the process group."""
    forced, plain = _bench(True, '--shared-model', '--backend', 'sweeps'), _bench(False, '--shared-model', '--backend', 'sweeps')
    assert forced['collective_backend'] == 'nccl' and plain.get('collective_backend') is None
    assert forced['per_rank'][0]['instances_sharing_factor'] == 64 == plain['per_rank'][0]['instances_sharing_factor']
    assert forced['mean_admm_iters'] == plain['mean_admm_iters'] and forced['solved_fraction_last_step'] == 1.0
    assert 'ONE' in forced['config']['workload']
