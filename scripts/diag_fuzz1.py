import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pympc_amd import BatchMPCController, fixtures
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
keys = ('x0', 'xref', 'uref', 'uminus1', 'Qx', 'QxN', 'Qu', 'QDu', 'xmin', 'xmax', 'umin', 'umax', 'Dumin', 'Dumax')
nx, nu, Np, Nc, B, soft = 4, 2, 10, 3, 4, True
for rep in range(3):
    rng = np.random.default_rng(52000 + seed)
    for _ in range(6): rng.random()
    kws = [fixtures.random_lti(53000 + 7 * seed + i, nx=nx, nu=nu, Np=Np, xbox=4.0) for i in range(B)]
    stack = lambda k: np.stack([np.asarray(kw[k], dtype=float) for kw in kws])
    def make():
        K = BatchMPCController(stack('Ad'), stack('Bd'), Np=Np, Nc=Nc, eps_feas=np.array([[kw.get('eps_feas', 1e6)] for kw in kws]), SOFT_ON=soft, **{k: stack(k) for k in keys})
        K.setup(); return K
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Kd, Ks = make(), make()
        w = 0.01 * np.random.default_rng(7).standard_normal((6, B, nx))
        tr = Kd.run(6, w=w)
        for k in range(6):
            u = Ks.output()
            infos0 = None
            print('rep', rep, 'step', k, 'u diff', np.abs(u - tr['u'][k]).max(), end=' ')
            Ks.update(tr['x'][k + 1])
            infos = Ks.prob.infos()
            print('status', [i.status for i in infos], list(tr['status'][k]), 'iter', [i.iter for i in infos], list(tr['iter'][k]))
