"""Multi-process (world_size 2, gloo, CPU) test of the instance sharding used by bench.py on N GPUs:
scatter of the problem data from rank 0, per-shard work, all-gather of u*.  The per-shard "solve" here is
the CPU oracle (tests may use it) so that the gathered result can be compared with a single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _solve_shard(Ad, Bd, x0):
    sys.path.insert(0, ROOT)
    from pympc_amd import MPCController, fixtures
    from oracle.osqp_oracle import OSQP
    us = []
    for i in range(Ad.shape[0]):
        kw = fixtures.random_lti(0, nx=5, nu=3, Np=8)
        kw.update(Ad=Ad[i].numpy(), Bd=Bd[i].numpy(), x0=x0[i].numpy(), eps_abs=1e-8, eps_rel=1e-8)
        K = MPCController(**kw); K.prob = OSQP(); K.setup()
        us.append(K.output())
    return torch.tensor(np.stack(us))


def _make(total):
    sys.path.insert(0, ROOT)
    from pympc_amd import fixtures
    kws = [fixtures.random_lti(40 + i, nx=5, nu=3, Np=8) for i in range(total)]
    return {k: torch.tensor(np.stack([kw[k] for kw in kws])) for k in ('Ad', 'Bd', 'x0')}


def _worker(rank, world_size, port, per, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    sys.path.insert(0, ROOT)
    from pympc_amd import sharding
    full = _make(per * world_size) if rank == 0 else None
    loc = sharding.scatter_instances(full, {'Ad': (5, 5), 'Bd': (5, 3), 'x0': (5,)}, per, torch.device('cpu'))
    lo, hi = sharding.shard_range(per * world_size, rank, world_size)
    assert (lo, hi) == (rank * per, (rank + 1) * per)
    u = _solve_shard(loc['Ad'], loc['Bd'], loc['x0'])
    u_all = sharding.gather_inputs(u)
    traj = torch.stack([u + 10.0 * k for k in range(4)])              # [steps, per, nu]: a device-loop launch's inputs
    tr_all = sharding.gather_trajectory(traj)
    assert tuple(tr_all.shape) == (world_size, 4, per, 3)
    assert torch.equal(tr_all[rank], traj) and torch.equal(tr_all[:, 0].reshape(-1, 3), u_all)
    if rank == 0:
        q.put(u_all.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_scatter_solve_gather_world2():
    per, ws = 3, 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, per, q)) for r in range(ws)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    full = _make(per * ws)
    ref = _solve_shard(full['Ad'], full['Bd'], full['x0']).numpy()
    assert got.shape == (per * ws, 3)
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-12)


def test_shard_range_handles_totals_that_do_not_divide():
    """Contiguous blocks of ceil(total / world) instances; the last rank(s) short or empty."""
    sys.path.insert(0, ROOT)
    from pympc_amd import sharding
    assert [sharding.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [sharding.shard_range(1024, r, 8) for r in range(8)] == [(128 * r, 128 * (r + 1)) for r in range(8)]
    assert [sharding.shard_range(5, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 5), (5, 5)]
    assert sharding.world() == (0, 1)


def _worker_uneven(rank, world_size, port, total, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    sys.path.insert(0, ROOT)
    from pympc_amd import sharding
    full = _make(total) if rank == 0 else None
    loc = sharding.scatter_instances(full, {'Ad': (5, 5), 'Bd': (5, 3), 'x0': (5,)}, None, torch.device('cpu'), total=total)
    lo, hi = sharding.shard_range(total, rank, world_size)
    assert loc['Ad'].shape == (hi - lo, 5, 5) and loc['x0'].shape == (hi - lo, 5)
    u = loc['x0'][:, :3] + loc['Ad'][:, 0, :3] * 2.0 + loc['Bd'][:, 4, :]        # something every array enters
    u_all = sharding.gather_inputs(u, total=total)
    traj = torch.stack([u + 10.0 * k for k in range(4)])
    tr_all = sharding.gather_trajectory(traj, total=total)
    assert tuple(u_all.shape) == (total, 3) and tuple(tr_all.shape) == (4, total, 3)
    assert torch.equal(tr_all[0], u_all) and torch.equal(tr_all[3], u_all + 30.0) and torch.equal(u_all[lo:hi], u)
    if rank == 0:
        q.put(u_all.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_packed_scatter_and_padded_gathers_with_a_short_last_rank_world3():
    """7 instances over 3 ranks (3, 3, 1): one packed scatter of three arrays, all-gathers padded to 3 rows and trimmed to 7."""
    total, ws = 7, 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_uneven, args=(r, ws, port, total, q)) for r in range(ws)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    full = _make(total)
    ref = (full['x0'][:, :3] + full['Ad'][:, 0, :3] * 2.0 + full['Bd'][:, 4, :]).numpy()
    assert np.array_equal(got, ref)


def _worker_shared(rank, world_size, port, total, q):
    """One model, many states over the ranks (last paragraph): broadcast the model once, scatter only x0, solve, gather u*."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    sys.path.insert(0, ROOT)
    from pympc_amd import sharding
    full = _make(total) if rank == 0 else None
    model = {k: full[k][0] for k in ('Ad', 'Bd')} if rank == 0 else None
    mdl = sharding.broadcast_model(model, {'Ad': (5, 5), 'Bd': (5, 3)}, torch.device('cpu'))
    loc = sharding.scatter_instances({'x0': full['x0']} if rank == 0 else None, {'x0': (5,)}, None, torch.device('cpu'), total=total)
    lo, hi = sharding.shard_range(total, rank, world_size)
    assert loc['x0'].shape == (hi - lo, 5) and mdl['Ad'].shape == (5, 5) and mdl['Bd'].shape == (5, 3)
    n = hi - lo
    u = _solve_shard(mdl['Ad'].expand(n, 5, 5), mdl['Bd'].expand(n, 5, 3), loc['x0']) if n else torch.zeros((0, 3), dtype=torch.float64)
    u_all = sharding.gather_inputs(u, total=total)
    if rank == 0:
        q.put((mdl['Ad'].numpy(), u_all.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_broadcast_model_scatter_states_world3():
    total, ws = 7, 3                                    # shards of 3, 3, 1
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_shared, args=(r, ws, port, total, q)) for r in range(ws)]
    for p in procs:
        p.start()
    Ad, got = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    full = _make(total)
    assert np.array_equal(Ad, full['Ad'][0].numpy())
    ref = _solve_shard(full['Ad'][:1].expand(total, 5, 5), full['Bd'][:1].expand(total, 5, 3), full['x0']).numpy()
    assert got.shape == (total, 3)
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-12)


def test_broadcast_model_without_a_process_group_is_a_copy():
    sys.path.insert(0, ROOT)
    from pympc_amd import sharding
    m = {'Ad': torch.arange(6.0, dtype=torch.float64).reshape(2, 3), 'c': torch.tensor(4.0, dtype=torch.float64)}
    out = sharding.broadcast_model(m, {'Ad': (2, 3), 'c': ()})
    assert torch.equal(out['Ad'], m['Ad']) and out['c'].shape == () and float(out['c']) == 4.0
