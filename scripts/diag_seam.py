import os, sys, warnings
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from util import load_golden, golden_csc, update_steps, golden_kwargs
from pympc_amd.solver import DeviceProblem
from pympc_amd import MPCController
from oracle.osqp_oracle import OSQP
for name in sys.argv[1:]:
    g = load_golden(name)
    P, A = golden_csc(g, 'P'), golden_csc(g, 'A')
    pd, po = DeviceProblem(), OSQP()
    pd.setup(P, g['q'], A, g['l'], g['u'], eps_abs=1e-3, eps_rel=1e-3); po.setup(P, g['q'], A, g['l'], g['u'], eps_abs=1e-3, eps_rel=1e-3)
    kw = golden_kwargs(g)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K = MPCController(**kw); K.setup(solve=False)
        for i, st in enumerate([None] + update_steps(g)):
            if st is not None:
                pd.update(l=st['l'], u=st['u_bound'], q=st['q']); po.update(l=st['l'], u=st['u_bound'], q=st['q'])
                K.update(st['x'], u=st['u'], xref=st['xref'], solve=False)
            rd, ro = pd.solve(), po.solve(); K.solve()
            print(name, 'step', i, 'raw-vs-oracle', np.abs(rd.x - ro.x).max(), 'mpc-vs-oracle', np.abs(K.res.x - ro.x).max(), 'iters', rd.info.iter, ro.info.iter, K.res.info.iter,
                  'obj', rd.info.obj_val - ro.info.obj_val, K.res.info.obj_val - ro.info.obj_val, 'max|x|', np.abs(ro.x).max())
            if i == 1 and 'upd0_output_u' in g.files:
                K.uminus1_rh = np.array(g['upd0_output_u'])
