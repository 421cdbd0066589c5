"""Read the controller data back out of a QP in the reference's layout (inverse of pyMPC/mpc.py:456-608).

The HIP solver works from (Ad, Bd, Qx, QxN, Qu, QDu, eps_feas) and applies P, A matrix-free; a caller that keeps the
reference's own builder and only swaps the solver object (``self.prob = osqp.OSQP()`` -> ``DeviceProblem()``,
mpc.py:241) hands over the assembled sparse ``P`` and ``A`` instead (mpc.py:266).  ``recover_model`` extracts the blocks
they were assembled from and -- the actual guarantee -- REBUILDS P and A from what it extracted and demands equality,
entry for entry (the last bit of the summed input-weight blocks excepted).  A QP that does not have pyMPC's structure is refused loudly; nothing is approximated.
"""
from types import SimpleNamespace

import numpy as np
import scipy.sparse as sp

from . import qp_build


class NotAnMPCQP(ValueError):
    pass


def _dims(P, A, nx=None, nu=None):
    n, m = P.shape[0], A.shape[0]
    Ac = sp.csc_matrix(A, copy=True)
    Ac.eliminate_zeros()                               # (scipy's kron keeps explicit zeros inside small dense blocks)
    per_col = np.diff(Ac.indptr)
    # the slack columns are the trailing columns with exactly one entry (their soft row); u columns have at least
    # three (input box, two Delta-u rows), x columns at least two (dynamics, soft row)
    n_x = 0
    while n_x < n and per_col[n - 1 - n_x] == 1:
        n_x += 1
    n_u, soft = n - 2 * n_x, True
    if n_x == 0:
        # SOFT_ON = False (mpc.py:237,530-597): no slack columns.  State box and input box are then one identity block over all n
        # columns, rows n_x .. n_x + n - 1; n_x is the first row R >= 1 at which such a block starts
        soft = False
        Ar = sp.csr_matrix(Ac)
        per_row, cols = np.diff(Ar.indptr), Ar.indices
        single = np.full(m, -1)
        one = per_row == 1
        single[one] = cols[Ar.indptr[:-1][one]]
        for R in range(1, m - n + 1):
            if np.array_equal(single[R:R + n], np.arange(n)):
                n_x = R
                break
        n_u = n - n_x
    if n_x < 2 or n_u < 1:
        raise NotAnMPCQP("A has neither pyMPC's block of slack columns nor (SOFT_ON = False) its identity block of box rows")
    if nu is None:
        nu = m - 2 * n_x - 2 * n_u                     # m = 2 n_x + n_u + (Nc+1) nu
    if nu < 1 or n_u % nu:
        raise NotAnMPCQP('row/column counts do not fit m = 2(Np+1)nx + Nc nu + (Nc+1) nu')
    if nx is None:
        # rows 0..nx-1 of the dynamics block hold only the -1 of x_0; row nx is the first with Ad/Bd entries
        per_row = np.diff(sp.csr_matrix(Ac[:n_x]).indptr)
        more = np.nonzero(per_row > 1)[0]
        nx = int(more[0]) if more.size else 0
    if nx < 1 or n_x % nx:
        raise NotAnMPCQP('could not determine nx from the dynamics rows')
    return int(nx), int(nu), n_x // nx - 1, n_u // nu, soft


def recover_model(P, A, l, u, nx=None, nu=None):
    """Returns ``dict(nx, nu, Np, Nc, Ad, Bd, Qx, QxN, Qu, QDu, eps_feas)`` of the controller whose QP matrices are
    exactly ``P`` (full symmetric or upper triangle) and ``A``; raises ``NotAnMPCQP`` otherwise.  ``l, u`` are checked
    for the stage-periodic structure of mpc.py:551-580 (the bound VALUES stay with the caller: they go to the solver
    verbatim through update_vectors)."""
    P, A = sp.csc_matrix(P), sp.csc_matrix(A)
    l, u = np.asarray(l, dtype=float), np.asarray(u, dtype=float)
    nx, nu, Np, Nc, soft = _dims(P, A, nx, nu)
    N, n_x, n_u = Np + 1, (Np + 1) * nx, Nc * nu
    n, m = (2 if soft else 1) * n_x + n_u, 2 * n_x + n_u + (Nc + 1) * nu
    if P.shape != (n, n) or A.shape != (m, n) or l.shape != (m,) or u.shape != (m,) or Np < 2:
        raise NotAnMPCQP('shapes do not fit an MPC QP with nx=%d nu=%d Np=%d Nc=%d' % (nx, nu, Np, Nc))
    Pu = sp.triu(P).tocsc()
    Pf = (Pu + sp.triu(Pu, 1).T).tocsc()               # what a solver that keeps triu(P) sees
    blk = lambda M, r, c, h, w: M[r:r + h, c:c + w].toarray()
    Ad, Bd = blk(A, nx, 0, nx, nx), blk(A, nx, n_x, nx, nu)
    Qx, QxN = blk(Pf, 0, 0, nx, nx), blk(Pf, Np * nx, Np * nx, nx, nx)
    D0 = blk(Pf, n_x, n_x, nu, nu)
    if Nc >= 2:
        QDu = -blk(Pf, n_x, n_x + nu, nu, nu)
        Qu = D0 - 2.0 * QDu
    else:                                              # one block iU Qu + QDu: any split gives the same P (q is the caller's)
        QDu, Qu = np.zeros((nu, nu)), D0 / float(Np)
    eps_feas = float(Pf[n_x + n_u, n_x + n_u]) if soft else 1e6      # (no slack block without soft constraints: the value is never used)
    model = dict(nx=nx, nu=nu, Np=Np, Nc=Nc, Ad=Ad, Bd=Bd, Qx=Qx, QxN=QxN, Qu=Qu, QDu=QDu, eps_feas=eps_feas, SOFT_ON=soft)

    # ---- the guarantee: rebuild and compare
    ctrl = SimpleNamespace(Np=Np, Nc=Nc, nx=nx, nu=nu, Ad=Ad, Bd=Bd, Qx=Qx, QxN=QxN, Qu=Qu, QDu=QDu, Qeps=eps_feas * sp.eye(nx),
                           xref=np.zeros(nx), uref=np.zeros(nu), uminus1=np.zeros(nu), x0=np.zeros(nx),
                           xmin=np.zeros(nx), xmax=np.zeros(nx), umin=np.zeros(nu), umax=np.zeros(nu), Dumin=np.zeros(nu), Dumax=np.zeros(nu),
                           JX_ON=True, JU_ON=True, JDU_ON=True, SOFT_ON=soft, COMPUTE_J_CNST=False)
    P2, _, A2, _, _, _, _ = qp_build.build_qp(ctrl)
    dA = (A - A2); dA.eliminate_zeros()
    P2u = sp.triu(P2).tocsc()
    dP = (Pu - P2u).tocsc(); dP.eliminate_zeros()
    # A must come back exactly.  P too, except that the input-weight blocks are stored as sums (Qu + 2 QDu, mpc.py:505-526):
    # splitting and re-adding them may move the last bit
    # ... and with Nc < Np the LAST input block is rebuilt as (Np - Nc + 1) Qu + QDu (mpc.py:513-517), which multiplies the rounding
    # error Qu = D0 - 2 QDu inherited from D0 by Np - Nc + 1: that block's bound is scaled by the same factor
    bad_P = False
    if dP.nnz:
        dPc = dP.tocoo()
        ptol = 4 * np.finfo(float).eps * max(1.0, np.abs(D0).max())
        last = (dPc.row >= n_x + (Nc - 1) * nu) & (dPc.col >= n_x + (Nc - 1) * nu)
        inside = (dPc.row >= n_x) & (dPc.row < n_x + n_u) & (dPc.col >= n_x) & (dPc.col < n_x + n_u)
        bad_P = bool((~inside).any() or (np.abs(dPc.data) > np.where(last, (Np - Nc + 1) * ptol, ptol)).any())
    if dA.nnz or bad_P:
        raise NotAnMPCQP('P, A are not the matrices pyMPC builds from their own blocks (%d / %d entries differ)' % (dP.nnz, dA.nnz))
    check_vectors(model, l, u)
    return model


def check_vectors(model, l=None, u=None):
    """l, u with the reference's structure (mpc.py:551-580, 404-408): equality rows l == u, zero behind the first nx;
    the state box, input box and Delta-u bounds repeat stage after stage (the first nu Delta-u rows carry + u_{-1})."""
    nx, nu, Np, Nc = model['nx'], model['nu'], model['Np'], model['Nc']
    n_x, n_u = (Np + 1) * nx, Nc * nu
    rs, ri, rdu = n_x, 2 * n_x, 2 * n_x + n_u
    for name, v in (('l', l), ('u', u)):
        if v is None:
            continue
        v = np.asarray(v, dtype=float)
        if v.shape != (rdu + (Nc + 1) * nu,):
            raise NotAnMPCQP('%s has the wrong length' % name)
        same = lambda block, period: np.array_equal(block.reshape(-1, period), np.broadcast_to(block[:period], (block.size // period, period)))
        if np.any(v[nx:n_x] != 0.0) or not same(v[rs:ri], nx) or not same(v[ri:rdu], nu) or not same(v[rdu + nu:], nu):
            raise NotAnMPCQP('%s does not have the stage-periodic structure of pyMPC/mpc.py:551-580' % name)
    if l is not None and u is not None and not np.array_equal(np.asarray(l)[:nx], np.asarray(u)[:nx]):
        raise NotAnMPCQP('the initial-state rows must be equalities (l[:nx] == u[:nx] = -x0)')
