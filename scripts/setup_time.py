import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pympc_amd.solver import BatchProblem
for (nx, nu, Np, B, xb) in [(12, 4, 30, 1024, 10.0), (20, 8, 100, 512, 1.0)]:
    d = bench.make_instances((nx, nu, Np, xb), 0, B)
    prob = BatchProblem(B, nx, nu, Np)
    eye = lambda k, s: np.broadcast_to(s * np.eye(k), (B, k, k))
    ones = lambda k, s: np.full((B, k), s)
    args = (d['Ad'], d['Bd'], eye(nx, 1.0), eye(nx, 1.0), eye(nu, .1), eye(nu, .1), ones(nx, -xb), ones(nx, xb), ones(nu, -1.), ones(nu, 1.),
            ones(nu, -.5), ones(nu, .5), ones(nu, 0.), np.full((B, 1), 1e6), d['x0'], ones(nu, 0.), np.zeros((B, nx)))
    prob.setup(*args)
    t = time.perf_counter(); prob.setup(*args); t1 = time.perf_counter() - t
    prob.profile(enable=True, reset=True)
    t = time.perf_counter(); prob.solve_async(); prob.synchronize(); t2 = time.perf_counter() - t
    it, chk, ref, sol = prob.stats(reset=True)
    print('(%d,%d,%d) x %d: setup %.1f ms (incl. uploads), cold solve %.1f ms, iters/solve %.1f, refactorizations/solve %.2f' % (nx, nu, Np, B, 1e3 * t1, 1e3 * t2, it / sol, ref / sol))
