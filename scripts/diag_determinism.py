"""Is a cold solve bit-reproducible from handle to handle? (development)"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pympc_amd import BatchMPCController, fixtures
keys = ('x0', 'xref', 'uref', 'uminus1', 'Qx', 'QxN', 'Qu', 'QDu', 'xmin', 'xmax', 'umin', 'umax', 'Dumin', 'Dumax')
def run(nx, nu, Np, Nc, B, seed=1, iters=None, first=0):
    kws = [fixtures.random_lti(53000 + 7 * seed + first + i, nx=nx, nu=nu, Np=Np, xbox=4.0) for i in range(B)]
    stack = lambda k: np.stack([np.asarray(kw[k], dtype=float) for kw in kws])
    sols = []
    for rep in range(3):
        K = BatchMPCController(stack('Ad'), stack('Bd'), Np=Np, Nc=Nc, eps_feas=np.array([[1e6]] * B), **{k: stack(k) for k in keys})
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            K.setup(solve=iters is None)
        if iters is not None:
            K.prob.iterate(iters)
            x = K.prob.iterate_state()[0]
        else:
            x = K.prob.solution()[0]
        D, E, c, rho = K.prob.scaling()
        sols.append((x.copy(), D.copy(), E.copy(), c.copy()))
        name = K.prob.kernel_name(False)
    d = [max(np.abs(sols[0][j] - sols[r][j]).max() for r in (1, 2)) for j in range(4)]
    print('(%d,%d,%d,%d) B=%d iters=%s %s: max diff between handles x %.2e D %.2e E %.2e c %.2e' % (nx, nu, Np, Nc, B, iters, name, d[0], d[1], d[2], d[3]))
for f in range(4):
    run(4, 2, 10, 3, 1, first=f)
for it in (25, 26, 50, 51, 75):
    run(4, 2, 10, 3, 4, iters=it)
from pympc_amd.solver import forced_settings; forced_settings(backend='sweeps').__enter__()      # (the block sweeps instead of the dense inverse)
run(4, 2, 10, 3, 4)
