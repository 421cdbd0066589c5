"""The MPC law without inequality constraints, as gain matrices -- the quantities of the reference's side script
test_scripts/alternative/unconstrained.py:170-183 (``k_x0, k_Xref, k_Uref, k_uminus1``) and doc/latex/main.tex:535-705, computed ON THE
DEVICE by the same block-tridiagonal KKT backend and ADMM kernels as every other solve.

Without inequality constraints the optimal input sequence is linear in (x0, xref, uref, u_{-1}):

    U* = K_x0 x0 + K_xref xref + K_uref uref + K_um1 u_{-1}          (constant reference xref; U* = (u_0 .. u_{Nc-1}) stacked)

so the columns of the gains are the solutions of the equality-constrained QP for the unit vectors of those arguments: one batch of
2 (nx + nu) instances of the controller with all bounds infinite, solved to a tight tolerance.  The reference builds the same gains by dense
condensing (prediction matrices, a Np nu x Np nu normal-equation solve); here nothing dense of that size is ever formed.

This is a utility next to the controller, not a shortcut inside it: ``MPCController.output()`` always reports what the ADMM solve of the
constrained QP returns (status, iteration count, iterate), like the reference's OSQP-backed class does."""
import warnings

import numpy as np

from .batch import BatchMPCController


def unconstrained_gains(Ad, Bd, Np, Nc=None, Qx=None, QxN=None, Qu=None, QDu=None, eps=1e-11, max_iter=400000, device=0):
    """Returns ``dict(K_x0 [Nc*nu, nx], K_xref [Nc*nu, nx], K_uref [Nc*nu, nu], K_um1 [Nc*nu, nu])``; the first nu rows are the
    feedback law of the receding-horizon controller, u_0 = K_x0[:nu] x0 + ...  Raises ``RuntimeError`` if a column does not converge."""
    Ad, Bd = np.asarray(Ad, dtype=float), np.asarray(Bd, dtype=float)
    nx, nu = Bd.shape
    Nc = Np if Nc is None else Nc
    B = 2 * (nx + nu)
    x0, xref, uref, um1 = np.zeros((B, nx)), np.zeros((B, nx)), np.zeros((B, nu)), np.zeros((B, nu))
    x0[np.arange(nx), np.arange(nx)] = 1.0
    xref[nx + np.arange(nx), np.arange(nx)] = 1.0
    uref[2 * nx + np.arange(nu), np.arange(nu)] = 1.0
    um1[2 * nx + nu + np.arange(nu), np.arange(nu)] = 1.0
    st = lambda M: None if M is None else np.broadcast_to(np.asarray(M, dtype=float), (B,) + np.asarray(M).shape)
    K = BatchMPCController(st(Ad), st(Bd), Np=Np, Nc=Nc, x0=x0, xref=xref, uref=uref, uminus1=um1,
                           Qx=st(Qx), QxN=st(QxN), Qu=st(Qu), QDu=st(QDu), eps_abs=eps, eps_rel=eps, device=device, max_iter=max_iter)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup()
        _, info = K.output(return_u_seq=True, return_status=True)
    if any(s != 'solved' for s in info['status']):
        raise RuntimeError('unconstrained_gains: a column did not converge: %r' % (sorted(set(info['status'])),))
    U = info['u_seq'].reshape(B, Nc * nu).T                      # column j = U* for the j-th unit argument
    return dict(K_x0=U[:, :nx].copy(), K_xref=U[:, nx:2 * nx].copy(), K_uref=U[:, 2 * nx:2 * nx + nu].copy(), K_um1=U[:, 2 * nx + nu:].copy())
