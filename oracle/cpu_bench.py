"""CPU baseline of bench.py (test infrastructure): the C oracle driven through the reference-shaped controller on the
BASELINE workload recipe -- one process per core, each walking its own instances through a warm-started receding
horizon.  Only osqp-equivalent work is timed (update(q,l,u) + solve); the numpy q/l/u refresh is not."""
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run_instances(args):
    first, count, steps, eps, nx, nu, Np, xbox, budget = args
    import numpy as np  # noqa: F401
    from pympc_amd import MPCController, fixtures, qp_build
    from oracle.osqp_oracle import OSQP
    t_solve, n_solve, iters, done = 0.0, 0, 0, 0
    t0 = time.perf_counter()
    for i in range(first, first + count):
        kw = fixtures.random_lti(i, nx=nx, nu=nu, Np=Np, xbox=xbox)
        kw.update(eps_abs=eps, eps_rel=eps)
        K = MPCController(**kw)
        K.prob = OSQP()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            K.setup()
        rng = fixtures.random_lti_noise_rng(i)
        x = kw['x0']
        for _ in range(steps):
            u = K.output()
            x = kw['Ad'] @ x + kw['Bd'] @ u + 0.01 * rng.standard_normal(nx)
            K.x0_rh, K.uminus1_rh = x, u
            q, _ = qp_build.refresh_vectors(K)
            ts = time.perf_counter()
            K.prob.update(q=q, l=K.l, u=K.u)
            K.res = K.prob.solve()
            t_solve += time.perf_counter() - ts
            n_solve += 1
            iters += K.res.info.iter
        done += 1
        if time.perf_counter() - t0 > budget:
            break
    return n_solve, t_solve, iters, done


def usable_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota (a container on a 256-thread host is
    often limited to a fraction of it; more workers than that only time-slice)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as g:
                quota, period = int(f.read()), int(g.read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return n


def all_cores(steps, eps, nx, nu, Np, xbox, budget, per_worker=64):
    """Every usable core of the box at once (spawned workers: nothing of the parent's GPU state is inherited)."""
    import multiprocessing as mp
    ncores = usable_cores()
    for v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):     # one thread per worker: no BLAS oversubscription
        os.environ[v] = '1'
    jobs = [(10000 + w * per_worker, per_worker, steps, eps, nx, nu, Np, xbox, budget) for w in range(ncores)]
    with mp.get_context('spawn').Pool(ncores) as pool:
        res = pool.map(run_instances, jobs)
    rate = sum(n / t for n, t, _, _ in res if t > 0)
    return dict(value=rate, cores=ncores, instances=sum(r[3] for r in res), cpu_seconds=sum(r[1] for r in res),
                mean_iters=sum(r[2] for r in res) / max(1, sum(r[0] for r in res)))
