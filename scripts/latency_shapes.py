"""MPCController.update() latency of ONE controller of several shapes (median / p95 over a closed loop) next to the CPU oracle on the
same loop, and which kernel served it (development / DESIGN.md "one controller of other shapes")."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pympc_amd import _lib
if os.environ.get('MPCQP_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['MPCQP_LIB'])
from pympc_amd import MPCController, fixtures


def loop(K, kw, nsim):
    x = np.asarray(kw['x0'], dtype=float).copy(); ts, its = [], []
    Ad, Bd = np.asarray(kw['Ad'], dtype=float), np.asarray(kw['Bd'], dtype=float).reshape(len(x), -1)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup()
        for i in range(nsim):
            u = K.output()
            x = Ad @ x + Bd @ u
            t = time.perf_counter(); K.update(x); ts.append(time.perf_counter() - t); its.append(K.res.info.iter)
    return 1e6 * np.array(ts), np.array(its)


CASES = [('cart_pole (4,1,20)', fixtures.cart_pole()), ('quadcopter (12,4,10)', fixtures.quadcopter()), ('random (12,4,30)', fixtures.random_lti(3)),
         ('random (8,2,20)', fixtures.random_lti(4, nx=8, nu=2, Np=20)), ('random (5,3,8)', fixtures.random_lti(6, nx=5, nu=3, Np=8)),
         ('random (10,4,25)', fixtures.random_lti(8, nx=10, nu=4, Np=25)), ('accel_brake', fixtures.accel_brake()),
         ('random (20,8,12)', fixtures.random_lti(5, nx=20, nu=8, Np=12)), ('point_mass_nc (2,1,25,10)', fixtures.point_mass_nc()),
         ('notebook (4,1,150,75)', dict(fixtures.cart_pole(), Np=150, Nc=75))]
want = sys.argv[1:]
for name, kw in CASES:
    if want and not any(w in name for w in want):
        continue
    nsim = 200
    K = MPCController(**kw)
    ts, its = loop(K, kw, nsim)
    from oracle.osqp_oracle import OSQP
    Ko = MPCController(**kw); Ko.prob = OSQP()
    tso, _ = loop(Ko, kw, nsim)
    print('%-28s %-34s GPU median %7.1f us p95 %7.1f | CPU oracle %7.1f us | %.1f its/step' % (name, K.prob.batch_problem.kernel_name(False), np.median(ts), np.percentile(ts, 95), np.median(tso), its.mean()))
