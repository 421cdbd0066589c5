import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, scipy.sparse as sp
from test_gpu_wide import _kw, _pair
from pympc_amd import qp_build
tag = sys.argv[1]
K, _ = _pair(_kw(tag)); K.setup(solve=False)
bp = K.prob.batch_problem
P, q, A, l, u = bp.export_qp()
Pr, qr, Ar, lr, ur = qp_build.build_qp(K)[:5]
U = sp.triu(Pr).toarray(); Pf = U + np.triu(U, 1).T
dP = np.argwhere(P[0] != Pf); dA = np.argwhere(A[0] != Ar.toarray())
print('n', bp.n, 'm', bp.m, 'P diffs', len(dP), dP[:5], 'A diffs', len(dA), dA[:8])
for r, c in dA[:5]: print('  A', r, c, A[0][r, c], Ar.toarray()[r, c])
for r, c in dP[:5]: print('  P', r, c, P[0][r, c], Pf[r, c])
