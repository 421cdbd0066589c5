"""Time K plain ADMM iterations on the cfg-3 batch with the library named by MPCQP_LIB (ablation builds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pympc_amd.solver import BatchProblem
B = int(os.environ.get('B', 1024)); iters = int(os.environ.get('ITERS', 100))
d = bench.make_instances(0, B)
prob = BatchProblem(B, 12, 4, 30)
eye = lambda k, s: np.broadcast_to(s * np.eye(k), (B, k, k))
ones = lambda k, s: np.full((B, k), s)
prob.setup(d['Ad'], d['Bd'], eye(12, 1.0), eye(12, 1.0), eye(4, .1), eye(4, .1), ones(12, -10.), ones(12, 10.), ones(4, -1.), ones(4, 1.),
           ones(4, -.5), ones(4, .5), ones(4, 0.), np.full((B, 1), 1e6), d['x0'], ones(4, 0.), np.zeros((B, 12)))
prob.iterate(10)
ts = []
for _ in range(5):
    t = time.perf_counter(); prob.iterate(iters); ts.append(time.perf_counter() - t)
print('%-40s B=%d  %.2f us/iteration (batch), min of 5' % (os.environ.get('MPCQP_LIB', 'default'), B, 1e6 * min(ts) / iters))
