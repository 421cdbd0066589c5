"""unconstrained gains: sweeps and time per call (development)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from pympc_amd import MPCController, fixtures
from test_gpu_unconstrained import _closed_form_gains
from pympc_amd import unconstrained as UU
if os.environ.get('RHO'): UU.RHO = float(os.environ['RHO'])
print('RHO', UU.RHO)
for name, kw in (('cart_pole', fixtures.cart_pole()), ('random_12_4_30', fixtures.random_lti(3)), ('quadcopter', fixtures.quadcopter()), ('point_mass_nc', fixtures.point_mass_nc()),
                 ('random_20_8_12', fixtures.random_lti(5, nx=20, nu=8, Np=12)), ('wide_40_8_10', fixtures.random_lti(7, nx=40, nu=8, Np=10)), ('notebook', dict(fixtures.cart_pole(), Np=150, Nc=75))):
    K = MPCController(**kw)
    try:
        G = K.unconstrained_gains()
    except Exception as e:
        print(name, 'FAILED', e); continue
    R = _closed_form_gains(kw)
    err = max(np.abs(G[k] - R[k]).max() / max(1.0, np.abs(R[k]).max()) for k in G)
    gs = K._gain_solver
    t0 = time.perf_counter()
    for _ in range(20): K.unconstrained_gains()
    t = (time.perf_counter() - t0) / 20
    print('%-16s sweeps %3d  err %.2e  residual %.1e  %.3f ms per call' % (name, gs.sweeps, err, gs.residuals.max(), 1e3 * t))
