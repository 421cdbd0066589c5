"""setup() alone of the headline batch, twice (development: kernel times of the setup launches under rocprofv3)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pympc_amd.solver import BatchProblem
nx, nu, Np, B, xb = 12, 4, 30, int(os.environ.get('B', 1024)), 10.0
d = bench.make_instances((nx, nu, Np, xb), 0, B)
prob = BatchProblem(B, nx, nu, Np)
eye = lambda k, s: np.broadcast_to(s * np.eye(k), (B, k, k))
ones = lambda k, s: np.full((B, k), s)
args = (d['Ad'], d['Bd'], eye(nx, 1.0), eye(nx, 1.0), eye(nu, .1), eye(nu, .1), ones(nx, -xb), ones(nx, xb), ones(nu, -1.), ones(nu, 1.),
        ones(nu, -.5), ones(nu, .5), ones(nu, 0.), np.full((B, 1), 1e6), d['x0'], ones(nu, 0.), np.zeros((B, nx)))
for _ in range(3):
    t = time.perf_counter(); prob.setup(*args); prob.synchronize(); print('setup %.2f ms' % (1e3 * (time.perf_counter() - t)))
