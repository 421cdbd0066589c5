"""Status / iteration histogram of the cold cfg-5 batch (512 x (20,8,100)) at a given tolerance (diagnostics)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from pympc_amd import fixtures
from test_gpu_loop_parity import _stacked_batch, _complete
B = int(os.environ.get('B', 512)); eps = float(os.environ.get('EPS', 1e-8))
kws = [_complete(fixtures.random_lti(i, nx=20, nu=8, Np=100, xbox=1.0)) for i in range(B)]
K = _stacked_batch(kws, eps_abs=eps, eps_rel=eps, max_iter=int(os.environ.get('MAXIT', 400000)))
K.setup()
infos = K.prob.infos()
st = collections.Counter(K.prob.status_string(i.status) for i in infos)
print('statuses', dict(st))
it = np.array([i.iter for i in infos]); ru = np.array([i.rho_updates for i in infos])
print('iters min/median/max', it.min(), np.median(it), it.max(), ' rho updates max', ru.max())
bad = [k for k, i in enumerate(infos) if i.status != 1]
for k in bad[:10]:
    i = infos[k]
    print('instance', k, K.prob.status_string(i.status), 'iter', i.iter, 'rho_upd', i.rho_updates, 'pri', i.pri_res, 'dua', i.dua_res, 'rho', i.rho, 'obj', i.obj_val)
