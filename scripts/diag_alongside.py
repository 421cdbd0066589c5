"""Device loop vs the oracle stepping alongside at the default tolerance, bench batches (development):
   python scripts/diag_alongside.py cfg5 [count]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from test_gpu_gaps import _bench_batch
from pympc_amd import MPCController
from oracle.osqp_oracle import OSQP
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg5'
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B, nx, nu, Np, xbox = (1024, 12, 4, 30, 10.0) if cfg == 'cfg3' else (512, 20, 8, 100, 1.0)
K, kws = _bench_batch(B, nx, nu, Np, xbox, 1e-3)
rng = np.random.default_rng(11)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    K.setup()
    inf0 = K.prob.infos()
    tr = K.run(50, w=0.01 * rng.standard_normal((50, B, nx)))
for i in np.unique(np.linspace(0, B - 1, count).astype(int)):
    kw = dict(kws[int(i)]); kw.update(eps_abs=1e-3, eps_rel=1e-3)
    Ko = MPCController(**kw); Ko.prob = OSQP()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Ko.setup()
        print('instance %d cold: gpu iter %d rho_upd %d rho %.6g | oracle iter %d rho_upd %s' % (i, inf0[int(i)].iter, inf0[int(i)].rho_updates, inf0[int(i)].rho, Ko.res.info.iter, getattr(Ko.res.info, 'rho_updates', '?')))
        for k in range(50):
            uo = Ko.output()
            du = np.abs(tr['u'][k, i] - uo).max() / max(1e-3, np.abs(uo).max())
            Ko.update(tr['x'][k + 1, i], tr['u'][k, i])
            if (Ko.res.info.iter, Ko.res.info.status_val) != (tr['iter'][k, i], tr['status'][k, i]) or du > 1e-7:
                print('   step %d: u rel diff %.2e  iter gpu %d oracle %d  status gpu %d oracle %d  oracle pri %.3e dua %.3e' % (k, du, tr['iter'][k, i], Ko.res.info.iter, tr['status'][k, i], Ko.res.info.status_val, getattr(Ko.res.info, 'pri_res', np.nan), getattr(Ko.res.info, 'dua_res', np.nan)))
