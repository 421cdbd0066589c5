"""Where a refactorization spends its time: MPCQP_LIB=<-DMPCQP_RUN_TIMING build> python scripts/diag_factor.py
(the timing build's mpcqp_get_stats prints thread 0's cycle split of factor_all: diag entries | off-diagonal entries | Mh = Ks Sn |
S -= Mh Ks' | Gauss-Jordan | store + tables)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pympc_amd import _lib
if os.environ.get('MPCQP_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MPCQP_LIB'])
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
    prob.refactor(); prob.synchronize(); prob.stats(reset=True)
    t = time.perf_counter()
    for _ in range(5):
        prob.refactor()
    prob.synchronize()
    print('(%d,%d,%d) x %d: %.3f ms per refactorization of the batch (%s)' % (nx, nu, Np, B, 1e3 * (time.perf_counter() - t) / 5, prob.kernel_name(loop=False)), file=sys.stderr)
    prob.stats(reset=True)
