"""Does mpcqp_refactor (run_factor_phase inside k_mpc_run) reproduce the factor k_setup made? (development)"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from pympc_amd import _lib
if os.environ.get('MPCQP_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['MPCQP_LIB'])
from pympc_amd import MPCController, fixtures
cases = {'notebook': dict(fixtures.cart_pole(), Np=150, Nc=75), 'nb_nc_eq': dict(fixtures.cart_pole(), Np=150, Nc=150), 'r8_2_60_20': dict(fixtures.random_lti(4, nx=8, nu=2, Np=60, xbox=4.0), Nc=20),
         'r8_2_60_60': fixtures.random_lti(4, nx=8, nu=2, Np=60, xbox=4.0), 'r12_4_40': fixtures.random_lti(4, nx=12, nu=4, Np=40, xbox=4.0), 'quadcopter_nc': dict(fixtures.quadcopter(), Nc=4),
         'r20_8_60': fixtures.random_lti(4, nx=20, nu=8, Np=60, xbox=4.0), 'r20_8_60_nc': dict(fixtures.random_lti(4, nx=20, nu=8, Np=60, xbox=4.0), Nc=20)}
for name, kw in cases.items():
    K = MPCController(**kw)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup(solve=False)
    bp = K.prob.batch_problem
    rhs = np.random.default_rng(1).standard_normal((1, bp.n))
    s0 = bp.kkt_solve(rhs)
    bp.refactor(); bp.synchronize()
    s1 = bp.kkt_solve(rhs)
    print('%-14s %-34s solve before/after refactor: max diff %.2e  nan %d' % (name, bp.kernel_name(False), np.nanmax(np.abs(s0 - s1)) if np.isfinite(s1).any() else np.nan, int(np.isnan(s1).sum())), flush=True)
