"""Time K plain ADMM iterations on the cfg-3 batch with the library named by MPCQP_LIB (ablation builds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pympc_amd import _lib
if os.environ.get('MPCQP_LIB'):            # ablation build: point the loader at it BEFORE the first load (scripts only)
    _lib.LIB_PATH = os.environ['MPCQP_LIB']
from pympc_amd.solver import BatchProblem
B = int(os.environ.get('B', 1024)); iters = int(os.environ.get('ITERS', 100))
NX, NU, NP = (int(os.environ.get(k, v)) for k, v in (('NX', 12), ('NU', 4), ('NP', 30)))
d = bench.make_instances((NX, NU, NP, float(os.environ.get('XBOX', 10.0))), 0, B)
prob = BatchProblem(B, NX, NU, NP)
eye = lambda k, s: np.broadcast_to(s * np.eye(k), (B, k, k))
ones = lambda k, s: np.full((B, k), s)
prob.setup(d['Ad'], d['Bd'], eye(NX, 1.0), eye(NX, 1.0), eye(NU, .1), eye(NU, .1), ones(NX, -10.), ones(NX, 10.), ones(NU, -1.), ones(NU, 1.),
           ones(NU, -.5), ones(NU, .5), ones(NU, 0.), np.full((B, 1), 1e6), d['x0'], ones(NU, 0.), np.zeros((B, NX)))
prob.iterate(10)
prob.stats(reset=True)
ts = []
for _ in range(5):
    t = time.perf_counter(); prob.iterate(iters); ts.append(time.perf_counter() - t)
print('%-40s B=%d  %.2f us/iteration (batch), min of 5' % (os.environ.get('MPCQP_LIB', 'default'), B, 1e6 * min(ts) / iters))
prob.stats()        # (a -DMPCQP_RUN_TIMING build prints its per-phase tick sums here: 100 MHz ticks summed over workgroups)
