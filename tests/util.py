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
