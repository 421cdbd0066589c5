"""what a refactorization (one rho update) costs one controller, by shape (development)"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pympc_amd import MPCController, fixtures
cases = {'notebook (4,1,150,75)': dict(fixtures.cart_pole(), Np=150, Nc=75), '(4,1,150,150)': dict(fixtures.cart_pole(), Np=150, Nc=150), 'cart_pole (4,1,20)': fixtures.cart_pole(),
         'quadcopter (12,4,10)': fixtures.quadcopter(), 'random (12,4,30)': fixtures.random_lti(3), 'random (8,2,60,20)': dict(fixtures.random_lti(4, nx=8, nu=2, Np=60, xbox=4.0), Nc=20)}
for name, kw in cases.items():
    K = MPCController(**kw)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup(solve=False)
    bp = K.prob.batch_problem
    bp.refactor(); bp.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): bp.refactor()
    bp.synchronize()
    print('%-24s %-34s refactorization %.1f us' % (name, bp.kernel_name(False), 1e6 * (time.perf_counter() - t0) / 20))
