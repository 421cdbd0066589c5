import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from test_gpu_group import _ctrl, SHAPES
os.environ['MPCQP_GROUP'] = sys.argv[1]
shape = SHAPES[int(sys.argv[2])]
eps = float(sys.argv[3])
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    K = _ctrl(shape, settings=dict(max_iter=400000), eps_abs=eps, eps_rel=eps)
    K.setup()
r = K.res
nx, nu, Np, Nc = shape[:4]
ou = (Np + 1) * nx; oe = ou + Nc * nu
print(shape, 'GROUP', sys.argv[1], 'eps', eps, r.info.status, r.info.iter, 'rho_upd', r.info.rho_updates, 'pri', r.info.pri_res, 'dua', r.info.dua_res,
      'nan x', np.isnan(r.x[:ou]).sum(), 'nan u', np.isnan(r.x[ou:oe]).sum(), 'nan eps', np.isnan(r.x[oe:]).sum(), 'nan y', np.isnan(r.y).sum())
u, info = K.output(return_u_seq=True)
print('  output u', u, 'nan in u_seq', np.isnan(info['u_seq']).sum())
