"""Wide stages (32 < nx + nu <= 64, mpcqp_wide.h): per-iteration and factorization time of a batch of random controllers, the CPU oracle
beside it.   python scripts/diag_wide.py [nx nu Np] [batch]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pympc_amd import _lib
if os.environ.get('MPCQP_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MPCQP_LIB'])        # e.g. a -DMPCQP_RUN_TIMING build: stats() then prints the phase split
from pympc_amd import BatchMPCController, MPCController, fixtures
nx, nu, Np = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (40, 8, 20)
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
kws = [fixtures.random_lti(1400 + i, nx=nx, nu=nu, Np=Np, xbox=3.0) for i in range(B)]
keys = ('x0', 'xref', 'uref', 'uminus1', 'Qx', 'QxN', 'Qu', 'QDu', 'xmin', 'xmax', 'umin', 'umax', 'Dumin', 'Dumax')
stack = lambda k: np.stack([np.asarray(kw[k], dtype=float) for kw in kws])
K = BatchMPCController(stack('Ad'), stack('Bd'), Np=Np, eps_feas=np.array([[kw.get('eps_feas', 1e6)] for kw in kws]), **{k: stack(k) for k in keys})
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    t0 = time.perf_counter(); K.setup(); t_setup = time.perf_counter() - t0
bp = K.prob
print('kernel', bp.kernel_name(loop=False), 'n', bp.n, 'm', bp.m, 'stream bytes/iter', bp.stream_bytes()[0])
res = {}
for iters in (50, 250):
    bp.iterate(iters)                                   # warm
    t0 = time.perf_counter()
    for _ in range(5):
        bp.iterate(iters)                               # (synchronous)
    res[iters] = (time.perf_counter() - t0) / 5
per_iter_us = 1e6 * (res[250] - res[50]) / 200
bp.stats(reset=True)
bp.refactor(); bp.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    bp.refactor()
bp.synchronize()
t_factor = (time.perf_counter() - t0) / 5
bp.stats(reset=True)
print('batch %d (%d,%d,%d): %.1f us per ADMM iteration, %.1f us per factorization, setup() %.1f ms; %.2f TB/s of factor stream' %
      (B, nx, nu, Np, per_iter_us, 1e6 * t_factor, 1e3 * t_setup, B * bp.stream_bytes()[0] / per_iter_us / 1e6))
from oracle.osqp_oracle import OSQP
Ko = MPCController(**kws[0]); Ko.prob = OSQP()
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    Ko.setup(solve=False)
    t0 = time.perf_counter(); Ko.prob.iterate(50); t50 = time.perf_counter() - t0
    t0 = time.perf_counter(); Ko.prob.iterate(250); t250 = time.perf_counter() - t0
print('CPU oracle, one instance: %.1f us per ADMM iteration' % (1e6 * (t250 - t50) / 200))
