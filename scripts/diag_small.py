"""Phase split of ONE small controller's warm closed loop (cart-pole by default) with a -DMPCQP_RUN_TIMING build:
MPCQP_LIB=<timing lib> [BACKEND=sweeps|dense|bcr|bcr8|bcrt] [NP= NC=] [NX= NU= NP= SEED= : fixtures.random_lti] python scripts/diag_small.py [fixture] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pympc_amd import _lib
if os.environ.get('MPCQP_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MPCQP_LIB'])
from pympc_amd import fixtures
from pympc_amd.solver import BatchProblem, forced_settings
if os.environ.get('BACKEND'):
    forced_settings(backend=os.environ['BACKEND']).__enter__()
name = sys.argv[1] if len(sys.argv) > 1 else 'cart_pole'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
kw = fixtures.random_lti(int(os.environ.get('SEED', '3')), nx=int(os.environ['NX']), nu=int(os.environ['NU']), Np=int(os.environ['NP']), xbox=float(os.environ.get('XBOX', '10'))) if os.environ.get('NX') else getattr(fixtures, name)()
if os.environ.get('NP'): kw = dict(kw, Np=int(os.environ['NP']), Nc=int(os.environ.get('NC', os.environ['NP'])))
nx, nu = np.asarray(kw['Bd']).reshape(np.asarray(kw['Ad']).shape[0], -1).shape
Np = kw['Np']; Nc = kw.get('Nc') or Np
one = lambda a, shp: np.asarray(a, dtype=float).reshape((1,) + shp)
bp = BatchProblem(1, nx, nu, Np, Nc)
xref = np.asarray(kw['xref'], dtype=float)
bp.setup(one(kw['Ad'], (nx, nx)), one(kw['Bd'], (nx, nu)), one(kw['Qx'], (nx, nx)), one(kw['QxN'], (nx, nx)), one(kw['Qu'], (nu, nu)), one(kw['QDu'], (nu, nu)),
         one(kw['xmin'], (nx,)), one(kw['xmax'], (nx,)), one(kw['umin'], (nu,)), one(kw['umax'], (nu,)), one(kw['Dumin'], (nu,)), one(kw['Dumax'], (nu,)),
         one(kw.get('uref', np.zeros(nu)), (nu,)), np.array([[kw.get('eps_feas', 1e6)]]), one(kw['x0'], (nx,)), one(kw.get('uminus1', np.zeros(nu)), (nu,)), xref.reshape(1, -1))
bp.solve_async(); u = bp.u0()
x = np.asarray(kw['x0'], dtype=float)
Ad, Bd = np.asarray(kw['Ad'], dtype=float), np.asarray(kw['Bd'], dtype=float).reshape(nx, nu)
for _ in range(20):
    x = Ad @ x + Bd @ u[0]; bp.update(x0=x[None]); bp.solve_async(); u = bp.u0()
print('kernel', bp.kernel_name(False), 'n', bp.n, 'm', bp.m, file=sys.stderr)
bp.stats(reset=True)
bp.profile(enable=True, reset=True)
t0 = time.perf_counter()
for _ in range(steps):
    x = Ad @ x + Bd @ u[0]; bp.update(x0=x[None]); bp.solve_async(); u = bp.u0()
t1 = time.perf_counter()
ms, nl = bp.profile(enable=False)
st = bp.stats()
print('%s: %.1f us/step host, kernel %.1f us, %.1f iterations/solve, %.2f rounds/solve' % (name, 1e6 * (t1 - t0) / steps, 1e3 * ms / max(1, nl), st[0] / max(1, st[3]), st[1] / max(1, st[3])))
