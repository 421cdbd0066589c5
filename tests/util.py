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


class KW(dict):
    """Constructor kwargs of a fixture; ``attrs`` = hidden switches to set on the controller afterwards (SOFT_ON, mpc.py:233-238)."""
    attrs = {}


def golden_kwargs(g):
    """Constructor kwargs stored in a golden file (keys ``in_*``), attribute switches (keys ``attr_*``) in ``.attrs``."""
    kw = KW()
    for k in g.files:
        if k.startswith('in_'):
            v = g[k]
            kw[k[3:]] = int(v) if k[3:] in ('Np', 'Nc') else (float(v) if v.ndim == 0 else np.array(v))
    kw.attrs = {k[5:]: bool(g[k]) for k in g.files if k.startswith('attr_')}
    return kw


def apply_attrs(K, kw):
    for k, v in getattr(kw, 'attrs', {}).items():
        setattr(K, k, v)
    return K


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


# ---- closed-loop golden trajectories (tests/golden/make_traj.py) ------------------------------------------------------
def traj_names(output_feedback=False):
    """State-feedback closed loops of pyMPC/mpc.py (default) or the output-feedback ones (traj_kalman_*.npz); the loop of
    the older mpc_no_slack.py class (traj_no_slack_*.npz) has a test of its own."""
    names = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, 'traj_*.npz')))
    return [n for n in names if n.startswith('kalman_') == output_feedback and not n.startswith('no_slack_')]


def load_traj(name):
    return np.load(os.path.join(GOLDEN_DIR, 'traj_%s.npz' % name), allow_pickle=False)


def cart_pole_plant(x, u, Ts=50e-3):
    """Nonlinear cart-pole + forward Euler, the plant of examples/example_inverted_pendulum.py:10-17,92-103
    (same restatement as in tests/golden/make_traj.py)."""
    M, m, b, ftheta, l, g = 0.5, 0.2, 0.1, 0.1, 0.3, 9.81
    F, v, theta, omega = float(u[0]), x[1], x[2], x[3]
    der = np.zeros(4)
    der[0] = v
    der[1] = (m * l * np.sin(theta) * omega ** 2 - m * g * np.sin(theta) * np.cos(theta) + m * ftheta * np.cos(theta) * omega + F - b * v) / (M + m * (1 - np.cos(theta) ** 2))
    der[2] = omega
    der[3] = ((M + m) * (g * np.sin(theta) - ftheta * omega) - m * l * omega ** 2 * np.sin(theta) * np.cos(theta) - (F - b * v) * np.cos(theta)) / (l * (M + m * (1 - np.cos(theta) ** 2)))
    return x + der * Ts
