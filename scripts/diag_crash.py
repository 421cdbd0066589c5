import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from pympc_amd import _lib
if os.environ.get('MPCQP_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['MPCQP_LIB'])
from pympc_amd import MPCController, fixtures
mode = sys.argv[1]
kw = dict(fixtures.cart_pole(), Np=int(os.environ.get('NP', 150)), Nc=int(os.environ.get('NC', 75)))
K = MPCController(**kw)
if mode == 'norho': K.solver_settings = dict(adaptive_rho=0)
if mode == 'iter99': K.solver_settings = dict(max_iter=99)
if mode == 'iter100': K.solver_settings = dict(max_iter=100)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    print('setup', flush=True); K.setup(solve=False); print('setup done', flush=True)
    bp = K.prob.batch_problem
    if mode == 'plain':
        bp.iterate(125); print('plain done', flush=True)
    else:
        K.solve(); print('solve done', K.res.info.status, K.res.info.iter, K.res.info.rho_updates, flush=True)
