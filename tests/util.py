"""Shared helpers for the test-suite: golden-vector loading and controller construction."""
import glob
import os

import numpy as np
import scipy.sparse as sp

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden_names(prefix='qp_'):
    return sorted(os.path.basename(p)[len(prefix):-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + '*.npz')))


def load_golden(name, prefix='qp_'):
    return np.load(os.path.join(GOLDEN_DIR, '%s%s.npz' % (prefix, name)))


def golden_kwargs(g):
    """Constructor kwargs stored in a golden file (keys ``in_*``)."""
    kw = {}
    for k in g.files:
        if k.startswith('in_'):
            v = g[k]
            kw[k[3:]] = int(v) if k[3:] in ('Np', 'Nc') else (float(v) if v.ndim == 0 else np.array(v))
    return kw


def golden_csc(g, which):
    return sp.csc_matrix((g[which + '_data'], g[which + '_indices'], g[which + '_indptr']),
                         shape=tuple(g[which + '_shape']))


def update_steps(g):
    steps = []
    s = 0
    while 'upd%d_x' % s in g.files:
        u = g['upd%d_u' % s]
        xr = g['upd%d_xref' % s]
        steps.append(dict(x=g['upd%d_x' % s], u=(u if u.size else None), xref=(xr if xr.size else None),
                          q=g['upd%d_q' % s], l=g['upd%d_l' % s], u_bound=g['upd%d_u_bound' % s]))
        s += 1
    return steps


def kkt_certificate(P, q, A, l, u, x, y):
    """(stationarity, primal violation, complementarity) in the inf-norm, relative to data size."""
    Ax = A @ x
    stat = np.abs(P @ x + q + A.T @ y).max() / max(1.0, np.abs(P @ x).max(), np.abs(q).max(), np.abs(A.T @ y).max())
    pv = max(0.0, (l - Ax).max(), (Ax - u).max()) / max(1.0, np.abs(Ax).max())
    yp, ym = np.maximum(y, 0), np.minimum(y, 0)
    ysc = max(1.0, np.abs(y).max())
    with np.errstate(invalid='ignore'):
        cu = np.where(np.isfinite(u), yp * (u - Ax), np.where(yp > 1e-9 * ysc, np.inf, 0.0)).max()
        cl = np.where(np.isfinite(l), -ym * (Ax - l), np.where(ym < -1e-9 * ysc, np.inf, 0.0)).max()
    return stat, pv, max(cu, cl) / ysc
