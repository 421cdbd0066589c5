#!/usr/bin/env python3
"""Optimum golden vectors: for every structural fixture (qp_<name>.npz, captured from the
reference) solve the reference-built QP to ~1e-11 residuals with the CPU oracle, CERTIFY the
result solver-independently with the KKT conditions evaluated on the reference-built matrices,
cross-check the small cases with the HiGHS QP solver bundled in scipy, and store
x*, y*, u0*, objective.  The MPC QP has a unique minimiser whenever Qu>0 or QDu>0
(SURVEY.md section 8c), so these vectors are solver-independent facts about the reference's QP.

    python tests/golden/make_optimum.py [name ...]  # writes tests/golden/opt_<name>.npz (all fixtures by default)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

from util import golden_names, load_golden, golden_csc, kkt_certificate  # noqa: E402
from oracle.osqp_oracle import OSQP                     # noqa: E402


def highs_qp(P, q, A, l, u, time_limit=60.0):
    """Independent QP solve with scipy's bundled HiGHS (active-set QP)."""
    from scipy.optimize._highspy import _core as hs
    import scipy.sparse as sp
    n, m = P.shape[0], A.shape[0]
    inf = hs.kHighsInf
    model = hs.HighsModel()
    lp = model.lp_
    lp.num_col_, lp.num_row_ = n, m
    lp.col_cost_ = np.asarray(q, dtype=float)
    lp.col_lower_ = -inf * np.ones(n)
    lp.col_upper_ = inf * np.ones(n)
    lp.row_lower_ = np.where(np.isfinite(l), l, -inf)
    lp.row_upper_ = np.where(np.isfinite(u), u, inf)
    Ac = sp.csc_matrix(A)
    lp.a_matrix_.format_ = hs.MatrixFormat.kColwise
    lp.a_matrix_.start_ = Ac.indptr.astype(np.int32)
    lp.a_matrix_.index_ = Ac.indices.astype(np.int32)
    lp.a_matrix_.value_ = Ac.data.astype(float)
    Pl = sp.tril(sp.csc_matrix(P), format='csc')
    hess = model.hessian_
    hess.dim_ = n
    hess.format_ = hs.HessianFormat.kTriangular
    hess.start_ = Pl.indptr.astype(np.int32)
    hess.index_ = Pl.indices.astype(np.int32)
    hess.value_ = Pl.data.astype(float)
    h = hs._Highs()
    h.setOptionValue('output_flag', False)
    h.setOptionValue('time_limit', float(time_limit))
    h.passModel(model)
    h.run()
    sol = h.getSolution()
    return np.array(sol.col_value), str(h.getModelStatus())


def main():
    for name in golden_names():
        if sys.argv[1:] and name not in sys.argv[1:]:
            continue
        g = load_golden(name)
        P, A = golden_csc(g, 'P'), golden_csc(g, 'A')
        q, l, u = g['q'], g['l'], g['u']
        prob = OSQP()
        prob.setup(P, q, A, l, u, eps_abs=1e-11, eps_rel=1e-11, max_iter=400000)
        r = prob.solve()
        assert r.info.status == 'solved', (name, r.info.status)
        stat, pv, comp = kkt_certificate(P, q, A, l, u, r.x, r.y)
        assert stat < 1e-9 and pv < 1e-9 and comp < 1e-9, (name, stat, pv, comp)
        nx = int(g['in_Ad'].shape[0]); nu = int(g['in_Bd'].shape[1]); Np = int(g['in_Np'])
        ou = (Np + 1) * nx
        out = dict(x=r.x, y=r.y, u0=r.x[ou:ou + nu], obj_val=r.info.obj_val, iters=r.info.iter,
                   kkt=np.array([stat, pv, comp]))
        line = '%-18s iter=%6d  KKT stat %.1e pv %.1e comp %.1e' % (name, r.info.iter, stat, pv, comp)
        if P.shape[0] <= 320:
            xh, status = highs_qp(P, q, A, l, u)
            du = np.abs(xh[ou:ou + nu] - out['u0']).max() / max(1e-12, np.abs(out['u0']).max())
            out['highs_u0'] = xh[ou:ou + nu]
            out['highs_rel_diff_u0'] = du
            line += '  HiGHS[%s] rel|u0 diff| %.1e' % (status.split('.')[-1], du)
        np.savez_compressed(os.path.join(HERE, 'opt_%s.npz' % name), **out)
        print(line)


if __name__ == '__main__':
    main()
