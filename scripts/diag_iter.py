"""Where do the GPU iterates leave the oracle's?  python scripts/diag_iter.py <golden fixture> [iters...]  (BACKEND=sweeps|dense|bcr|bcr8|bcrt in the environment of THIS script forces the backend through mpcqp_settings.backend)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from util import load_golden, golden_kwargs, apply_attrs
from pympc_amd import MPCController
from pympc_amd.solver import forced_settings
if os.environ.get('BACKEND'):
    forced_settings(backend=os.environ['BACKEND']).__enter__()
from oracle.osqp_oracle import OSQP
name = sys.argv[1]; its = [int(v) for v in sys.argv[2:]] or [1, 2, 3]
kw = golden_kwargs(load_golden(name))
nx = np.asarray(kw['Ad']).shape[0]; nu = np.asarray(kw['Bd']).reshape(nx, -1).shape[1]; Np = kw['Np']; N = Np + 1
for k in its:
    K = apply_attrs(MPCController(**kw), kw); K.setup(solve=False)
    Ko = apply_attrs(MPCController(**kw), kw); Ko.prob = OSQP(); Ko.setup(solve=False)
    print(K.prob.batch_problem.kernel_name(False))
    K.prob.batch_problem.iterate(k); Ko.prob.iterate(k)
    x, z, y = (v[0] for v in K.prob.batch_problem.iterate_state()); xo, zo, yo, _ = Ko.prob.iterate_state()
    nxs, nus = N * nx, Np * nu
    blocks = {'x': slice(0, nxs), 'u': slice(nxs, nxs + nus), 'eps': slice(nxs + nus, None)}
    rows = {'dyn': slice(0, nxs), 'sbox': slice(nxs, 2 * nxs), 'ubox': slice(2 * nxs, 2 * nxs + nus), 'du0': slice(2 * nxs + nus, 2 * nxs + nus + nu), 'du': slice(2 * nxs + nus + nu, None)}
    print('iters', k, {b: float(np.abs(x[s] - xo[s]).max()) for b, s in blocks.items()})
    print('   z', {b: float(np.abs(z[s] - zo[s]).max()) for b, s in rows.items()})
    print('   y', {b: float(np.abs(y[s] - yo[s]).max()) for b, s in rows.items()})
    if k == its[-1]:
        d = np.abs(x[blocks['u']] - xo[blocks['u']]).reshape(Np, nu); print('   u err per stage (max over inputs):', np.round(d.max(axis=1), 5))
        d = np.abs(x[blocks['x']] - xo[blocks['x']]).reshape(N, nx); print('   x err per stage:', np.round(d.max(axis=1), 5))
