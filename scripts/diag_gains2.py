import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pympc_amd import fixtures
from pympc_amd import unconstrained as U
name = sys.argv[1] if len(sys.argv) > 1 else 'cart_pole'
kw = fixtures.random_lti(3) if name == 'random' else getattr(fixtures, name)()
nx, nu = np.asarray(kw['Bd']).reshape(np.asarray(kw['Ad']).shape[0], -1).shape
for rho in (100.0, 1.0):
    U.RHO = rho
    gs = U.GainSolver(nx, nu, kw['Np'], kw.get('Nc') or kw['Np'])
    try:
        gs.gains(kw['Ad'], np.asarray(kw['Bd']).reshape(nx, nu), kw['Qx'], kw.get('QxN', kw['Qx']), kw['Qu'], kw['QDu'])
    except Exception as e:
        print('rho', rho, e)
    print(gs.prob.kernel_name(False))
    for j, i in enumerate(gs.prob.infos()):
        print('  col %2d status %d iter %3d pri %.2e dua %.2e rho %.3g' % (j, i.status, i.iter, i.pri_res, i.dua_res, i.rho))
