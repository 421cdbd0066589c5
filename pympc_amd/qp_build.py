"""Host-side (numpy/scipy) construction of the MPC quadratic program

    min 1/2 w'Pw + q'w   s.t.  l <= A w <= u,     w = [x_0..x_Np | u_0..u_{Nc-1} | eps_0..eps_Np]

exactly as the reference hands it to its solver (pyMPC/mpc.py:456-608 for the full build,
pyMPC/mpc.py:386-454 for the per-step q/l/u refresh).  This module exists so that the
drop-in ``MPCController`` can expose the same public ``P, q, A, l, u`` attributes; the
numbers the solver works on are built independently on the GPU by ``csrc/mpcqp*.h`` and
the two are compared in the tests.

Layout facts reproduced bug-for-bug:

* rows: dynamics ``(Np+1)nx`` | soft state box ``(Np+1)nx`` | input box ``Nc nu`` | Delta-u ``(Nc+1)nu``;
* the Delta-u difference rows use a super-diagonal offset of ONE SCALAR of the flattened input
  sequence (mpc.py:570), so for nu>1 they couple neighbouring channels and the last row is
  ``-u_flat[-1]``;
* the stored sparsity pattern (including the explicit zeros scipy keeps when ``kron`` takes its
  dense-block route) follows scipy 1.15's behaviour for the reference's expression tree.
"""
import numpy as np
import scipy.sparse as sp


def _dense(M):
    return M.toarray() if sp.issparse(M) else np.asarray(M, dtype=float)


def _stored_mask(B):
    """Entries of a kron right-operand that end up stored: scipy.sparse.kron keeps a whole
    dense block (explicit zeros included) when 2*nnz >= size, otherwise only the nonzeros."""
    Bd = _dense(B)
    nnz = B.nnz if sp.issparse(B) else int(np.count_nonzero(Bd))
    if nnz == 0:
        return np.zeros(Bd.shape, dtype=bool), False
    if 2 * nnz >= Bd.size:
        return np.ones(Bd.shape, dtype=bool), True
    if sp.issparse(B):
        m = np.zeros(Bd.shape, dtype=bool)
        Bc = B.tocoo()
        m[Bc.row, Bc.col] = True
        return m, False
    return Bd != 0, False


class _Triplets:
    """Accumulates COO triplets of one matrix."""

    def __init__(self):
        self.r, self.c, self.v = [], [], []

    def block(self, r0, c0, M, mask=None):
        M = np.asarray(M, dtype=float)
        if mask is None:
            mask = M != 0
        ii, jj = np.nonzero(mask)
        self.r.append(ii + r0)
        self.c.append(jj + c0)
        self.v.append(M[ii, jj])

    def diag(self, r0, c0, vals):
        vals = np.asarray(vals, dtype=float)
        k = np.arange(vals.size)
        self.r.append(k + r0)
        self.c.append(k + c0)
        self.v.append(vals)

    def csc(self, shape):
        if not self.r:
            return sp.csc_matrix(shape)
        r = np.concatenate(self.r)
        c = np.concatenate(self.c)
        v = np.concatenate(self.v)
        return sp.coo_matrix((v, (r, c)), shape=shape).tocsc()


def horizon_weights(Np, Nc):
    """iU of mpc.py:514-515 (last input held for Np-Nc+1 steps) and the tridiagonal
    Delta-u coupling iDu of mpc.py:522-523."""
    iU = np.ones(Nc)
    iU[Nc - 1] = Np - Nc + 1
    iDu = 2 * np.eye(Nc) - np.eye(Nc, k=1) - np.eye(Nc, k=-1)
    iDu[Nc - 1, Nc - 1] = 1
    return iU, iDu


def state_cost_matrix(ctrl):
    """P_X = blkdiag(I_Np (x) Qx, QxN) with only nonzero values stored (mpc.py:482-487)."""
    Np, nx = ctrl.Np, ctrl.nx
    t = _Triplets()
    if ctrl.JX_ON:
        Qx, QxN = _dense(ctrl.Qx), _dense(ctrl.QxN)
        for k in range(Np):
            t.block(k * nx, k * nx, Qx)
        t.block(Np * nx, Np * nx, QxN)
    return t.csc(((Np + 1) * nx, (Np + 1) * nx))


def linear_cost(ctrl, P_X, uminus1):
    """q and the constant J_CNST for the controller's current xref/uref and the given u_{-1}
    (mpc.py:489-526 at setup, mpc.py:411-452 per step; both use the same formulas)."""
    Np, Nc, nx, nu = ctrl.Np, ctrl.Nc, ctrl.nx, ctrl.nu
    xref, uref = ctrl.xref, ctrl.uref
    q_X = np.zeros((Np + 1) * nx)
    J = 0.0
    if ctrl.JX_ON:
        if xref.ndim == 2 and xref.shape[0] >= Np + 1:
            q_X += (-xref.reshape(1, -1) @ P_X).ravel()
            if ctrl.COMPUTE_J_CNST:
                J += -1 / 2 * q_X @ xref.ravel()
        else:
            q_X += np.hstack([np.kron(np.ones(Np), -ctrl.Qx.dot(xref)), -ctrl.QxN.dot(xref)])
            if ctrl.COMPUTE_J_CNST:
                # the reference weighs all Np+1 terms with QxN here (mpc.py:500)
                J += 1 / 2 * Np * (xref.dot(ctrl.QxN.dot(xref))) + 1 / 2 * xref.dot(ctrl.QxN.dot(xref))
    q_U = np.zeros(Nc * nu)
    if ctrl.JU_ON:
        J += 1 / 2 * Np * (uref.dot(ctrl.Qu.dot(uref)))
        if Nc == Np:
            q_U += np.kron(np.ones(Nc), -ctrl.Qu.dot(uref))
        else:
            iU, _ = horizon_weights(Np, Nc)
            q_U += np.kron(iU, -ctrl.Qu.dot(uref))
    if ctrl.JDU_ON:
        J += 1 / 2 * uminus1.dot((ctrl.QDu).dot(uminus1))
        q_U += np.hstack([-ctrl.QDu.dot(uminus1), np.zeros((Nc - 1) * nu)])
    if ctrl.SOFT_ON:
        q = np.hstack([q_X, q_U, np.zeros((Np + 1) * nx)])
    else:
        q = np.hstack([q_X, q_U])
    return q, J


def build_qp(ctrl):
    """Full QP build.  Returns ``(P, q, A, l, u, P_X, J_CNST)`` with P, A in CSC."""
    Np, Nc, nx, nu = ctrl.Np, ctrl.Nc, ctrl.nx, ctrl.nu
    N = Np + 1
    n_x, n_u = N * nx, Nc * nu
    n_eps = n_x if ctrl.SOFT_ON else 0
    n = n_x + n_u + n_eps

    # ---- cost -------------------------------------------------------------------------
    P_X = state_cost_matrix(ctrl)
    iU, iDu = horizon_weights(Np, Nc)
    tP = _Triplets()
    Pc = P_X.tocoo()
    tP.r.append(Pc.row.astype(np.int64)); tP.c.append(Pc.col.astype(np.int64)); tP.v.append(Pc.data)
    Qu, QDu = _dense(ctrl.Qu), _dense(ctrl.QDu)
    for k in range(Nc):
        for kk in (k - 1, k, k + 1):
            if kk < 0 or kk >= Nc:
                continue
            blk = np.zeros((nu, nu))
            if ctrl.JU_ON and kk == k:
                blk = blk + (iU[k] if Nc != Np else 1.0) * Qu
            if ctrl.JDU_ON and iDu[k, kk] != 0:
                blk = blk + iDu[k, kk] * QDu
            tP.block(n_x + k * nu, n_x + kk * nu, blk)
    if ctrl.SOFT_ON:
        Qeps = ctrl.Qeps
        mask, _ = _stored_mask(Qeps)
        Qe = _dense(Qeps)
        for k in range(N):
            tP.block(n_x + n_u + k * nx, n_x + n_u + k * nx, Qe, mask)
    P = tP.csc((n, n))

    uminus1 = ctrl.uminus1
    saved = ctrl.xref, ctrl.uref
    q, J = linear_cost(ctrl, P_X, uminus1)

    # ---- constraints ------------------------------------------------------------------
    tA = _Triplets()
    Ad, Bd = ctrl.Ad, ctrl.Bd
    Add, Bdd = _dense(Ad), _dense(Bd)
    # dynamics rows: -x_k + Ad x_{k-1} + Bd u_{min(k-1, Nc-1)} = (k == 0 ? -x0 : 0)
    _, minus_eye_dense = _stored_mask(-sp.eye(nx))
    ad_mask, _ = _stored_mask(Ad)
    if minus_eye_dense:                 # scipy adds two BSR operands block-wise: whole blocks stay
        eye_mask = np.ones((nx, nx), dtype=bool)
        ad_mask = np.ones((nx, nx), dtype=bool) if np.any(Add != 0) else ad_mask
    else:                               # element-wise sum: zero results are dropped
        eye_mask = np.eye(nx, dtype=bool)
        ad_mask = Add != 0
    for k in range(N):
        tA.block(k * nx, k * nx, -np.eye(nx), eye_mask)
        if k > 0:
            tA.block(k * nx, (k - 1) * nx, Add, ad_mask)
    bd_mask, _ = _stored_mask(Bd)
    for k in range(1, N):
        tA.block(k * nx, n_x + min(k - 1, Nc - 1) * nu, Bdd, bd_mask)
    r0 = n_x
    # state box on x_k (+ eps_k)
    tA.diag(r0, 0, np.ones(n_x))
    if ctrl.SOFT_ON:
        tA.diag(r0, n_x + n_u, np.ones(n_x))
    r0 += n_x
    # input box
    tA.diag(r0, n_x, np.ones(n_u))
    r0 += n_u
    # Delta-u rows: first nu rows pick u_0, then -I + superdiagonal(offset 1 scalar)
    tA.diag(r0, n_x, np.ones(nu))
    r0 += nu
    tA.diag(r0, n_x, -np.ones(n_u))
    if n_u > 1:
        tA.diag(r0, n_x + 1, np.ones(n_u - 1))
    m = r0 + n_u
    A = tA.csc((m, n))

    x0 = ctrl.x0
    leq = np.hstack([-x0, np.zeros(Np * nx)])
    l = np.hstack([leq, np.kron(np.ones(N), ctrl.xmin), np.kron(np.ones(Nc), ctrl.umin)])
    u = np.hstack([leq, np.kron(np.ones(N), ctrl.xmax), np.kron(np.ones(Nc), ctrl.umax)])
    ldu = np.kron(np.ones(Nc + 1), ctrl.Dumin)
    udu = np.kron(np.ones(Nc + 1), ctrl.Dumax)
    ldu[0:nu] += uminus1[0:nu]
    udu[0:nu] += uminus1[0:nu]
    l = np.hstack([l, ldu])
    u = np.hstack([u, udu])
    ctrl.xref, ctrl.uref = saved
    return P, q, A, l, u, P_X, J


def refresh_vectors(ctrl):
    """Per-step refresh of q, l, u in place of mpc.py:386-454 (x0_rh, uminus1_rh, xref may have changed)."""
    Np, Nc, nx, nu = ctrl.Np, ctrl.Nc, ctrl.nx, ctrl.nu
    ctrl.l[:nx] = -ctrl.x0_rh
    ctrl.u[:nx] = -ctrl.x0_rh
    off = 2 * (Np + 1) * nx + Nc * nu
    um1 = ctrl.uminus1_rh
    ctrl.l[off:off + nu] = ctrl.Dumin + um1[0:nu]
    ctrl.u[off:off + nu] = ctrl.Dumax + um1[0:nu]
    q, J = linear_cost(ctrl, ctrl.P_X, um1)
    return q, J
