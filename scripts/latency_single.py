"""Single-controller step latency (BASELINE configs[1]: cart-pole nx=4 nu=1 Np=20, one QP on one GPU) through the drop-in
MPCController API, next to the CPU oracle on the same loop."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pympc_amd import MPCController, fixtures

def loop(K, kw, nsim):
    x = kw['x0'].copy(); ts = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup()
        for i in range(nsim):
            u = K.output()
            x = kw['Ad'] @ x + kw['Bd'] @ u
            t = time.perf_counter(); K.update(x); ts.append(time.perf_counter() - t)
    return np.array(ts), K

kw = fixtures.cart_pole()
ts, K = loop(MPCController(**kw), kw, 400)
print('GPU  drop-in MPCController.update(): median %.1f us, p95 %.1f us  (iters last %d)' % (1e6 * np.median(ts), 1e6 * np.percentile(ts, 95), K.res.info.iter))
bp = K.prob.batch_problem
bp.profile(enable=True, reset=True)
x = K.x0_rh
t0 = time.perf_counter()
for i in range(200):
    bp.update(x0=x[None, :]); bp.solve_async(); u0 = bp.u0()
t1 = time.perf_counter()
ms, n = bp.profile()
print('     raw C ABI update+solve+get_u0: %.1f us per step, kernel %.1f us' % (1e6 * (t1 - t0) / 200, 1e3 * ms / max(1, n)))
from oracle.osqp_oracle import OSQP
Ko = MPCController(**kw); Ko.prob = OSQP()
ts, _ = loop(Ko, kw, 400)
print('CPU  oracle (1 core)              : median %.1f us' % (1e6 * np.median(ts)))
