"""error of the gains against the condensed closed form after each multiplier sweep, for several rho (development)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from pympc_amd import fixtures
from pympc_amd import unconstrained as U
from test_gpu_unconstrained import _closed_form_gains
cases = (('cart_pole', fixtures.cart_pole()), ('random_12_4_30', fixtures.random_lti(3)), ('quadcopter', fixtures.quadcopter()), ('point_mass_nc', fixtures.point_mass_nc()),
         ('random_20_8_12', fixtures.random_lti(5, nx=20, nu=8, Np=12)), ('wide_40_8_10', fixtures.random_lti(7, nx=40, nu=8, Np=10)))
for name, kw in cases:
    nx, nu = np.asarray(kw['Bd']).reshape(np.asarray(kw['Ad']).shape[0], -1).shape
    Np, Nc = kw['Np'], kw.get('Nc') or kw['Np']
    R = _closed_form_gains(kw)
    Rm = np.hstack([R['K_x0'], R['K_xref'], R['K_uref'], R['K_um1']])
    for rho in (1.0, 10.0, 100.0):
        U.RHO = rho
        gs = U.GainSolver(nx, nu, Np, Nc, tol=1e300)
        U.SWEEPS = 0
        try:
            gs.gains(kw['Ad'], np.asarray(kw['Bd']).reshape(nx, nu), kw['Qx'], kw.get('QxN', kw['Qx']), kw['Qu'], kw['QDu'])
        except Exception as e:
            pass
        line = []
        for k in range(1, 15):
            res = gs.prob.eq_solve(1, cold=(k == 1))
            x, _, _ = gs.prob.solution(want_y=False)
            ou = (Np + 1) * nx
            Um = x[:, ou:ou + Nc * nu].T
            err = np.abs(Um - Rm).max() / max(1.0, np.abs(Rm).max())
            rr = np.maximum(res[:, 0] / np.maximum(1.0, res[:, 1]), res[:, 2] / np.maximum(1.0, res[:, 3])).max()
            line.append('%.0e/%.0e' % (err, rr))
        print('%-15s rho %5.0f err/res per sweep: %s' % (name, rho, ' '.join(line)))
