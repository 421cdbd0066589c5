import sys, warnings
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from util import *
from pympc_amd import MPCController
name = sys.argv[1] if len(sys.argv)>1 else 'accel_brake'
g = load_golden(name)
K = MPCController(**golden_kwargs(g)); K.setup(solve=False)
bp = K.prob.batch_problem
for s, st in enumerate(update_steps(g)):
    K.update(st['x'], u=st['u'], xref=st['xref'], solve=False)
    _, q, _, l, u = bp.export_qp()
    d = np.abs(q[0]-st['q']); i = d.argmax()
    print(s, 'max diff', d.max(), 'at', i, q[0][i], st['q'][i], 'host q', K.q[i], 'um1', K.uminus1_rh, 'ou', (K.Np+1)*K.nx)
    print(' l diff', np.abs(l[0]-np.clip(st['l'],-1e30,1e30)).max(), 'u diff', np.abs(u[0]-np.clip(st['u_bound'],-1e30,1e30)).max())
