"""Solver-independent oracle for MPC problems in which no inequality constraint is active: the CONDENSED formulation.

With x_k = Ad^k x0 + sum_j Ad^(k-1-j) Bd u_j the states are eliminated and the cost of mpc.py:482-532 becomes an
unconstrained quadratic in the input sequence U, minimised in closed form -- the route of the reference's side script
test_scripts/alternative/unconstrained.py:141-183 (prediction matrices A_cal, B_cal; gains k_x0, k_Xref, k_Uref,
k_uminus1) and of doc/latex/main.tex:535-705.  Restated here with numpy only (test infrastructure), generalised to a
control horizon Nc < Np (held last input, mpc.py:513-517,540-543) and to a time-varying reference.  Nothing in it
shares code or method with the ADMM solvers (GPU library, oracle/osqp_ref.c): dense normal equations, one Cholesky.
"""
import numpy as np


def prediction_matrices(Ad, Bd, Np):
    """calA [Np*nx, nx] (rows Ad^1..Ad^Np) and calB [Np*nx, Np*nu] (block (k, j) = Ad^(k-j) Bd for j <= k):
    X = calA x0 + calB U for X = (x_1..x_Np), U = (u_0..u_{Np-1})."""
    nx, nu = Bd.shape
    powers = [np.eye(nx)]
    for _ in range(Np):
        powers.append(Ad @ powers[-1])
    calA = np.vstack(powers[1:])
    calB = np.zeros((Np * nx, Np * nu))
    for k in range(Np):
        for j in range(k + 1):
            calB[k * nx:(k + 1) * nx, j * nu:(j + 1) * nu] = powers[k - j] @ Bd
    return calA, calB


def unconstrained_mpc(Ad, Bd, Np, Nc=None, x0=None, xref=None, uref=None, uminus1=None, Qx=None, QxN=None, Qu=None, QDu=None, **_):
    """Minimiser of the MPC cost without inequality constraints.  Returns ``(u_seq [Nc, nu], x_seq [Np+1, nx])``.
    Extra keyword arguments (bounds, eps_feas, tolerances) are accepted and ignored so that a controller kwargs dict
    can be passed as is."""
    Ad, Bd = np.asarray(Ad, dtype=float), np.asarray(Bd, dtype=float)
    nx, nu = Bd.shape
    Nc = Np if Nc is None else Nc
    z = lambda v, k: np.zeros(k) if v is None else np.asarray(v, dtype=float)
    x0, uref, um1 = z(x0, nx), z(uref, nu), z(uminus1 if uminus1 is not None else uref, nu)
    Qx = np.zeros((nx, nx)) if Qx is None else np.asarray(Qx, dtype=float)
    QxN = Qx if QxN is None else np.asarray(QxN, dtype=float)
    Qu = np.zeros((nu, nu)) if Qu is None else np.asarray(Qu, dtype=float)
    QDu = np.zeros((nu, nu)) if QDu is None else np.asarray(QDu, dtype=float)
    xref = np.zeros(nx) if xref is None else np.asarray(xref, dtype=float)
    Xref = (xref[1:Np + 1] if xref.ndim == 2 else np.tile(xref, (Np, 1))).ravel()          # references of x_1..x_Np

    calA, calB = prediction_matrices(Ad, Bd, Np)
    hold = np.zeros((Np, Nc)); hold[np.arange(Np), np.minimum(np.arange(Np), Nc - 1)] = 1.0   # u_k = v_min(k, Nc-1)
    Bc = calB @ np.kron(hold, np.eye(nu))
    calQ = np.kron(np.eye(Np), Qx); calQ[-nx:, -nx:] = QxN                                  # x_1..x_{Np-1}: Qx, x_Np: QxN
    times_held = hold.sum(axis=0)                                                           # 1, .., 1, Np-Nc+1
    diff = 2 * np.eye(Nc) - np.eye(Nc, k=1) - np.eye(Nc, k=-1); diff[-1, -1] = 1.0          # sum_k |v_k - v_{k-1}|^2
    H = Bc.T @ calQ @ Bc + np.kron(np.diag(times_held), Qu) + np.kron(diff, QDu)
    g = Bc.T @ calQ @ (calA @ x0 - Xref) - np.kron(times_held, Qu @ uref)
    g[:nu] -= QDu @ um1
    V = np.linalg.solve(H, -g)
    X = np.concatenate([x0, calA @ x0 + Bc @ V]).reshape(Np + 1, nx)
    return V.reshape(Nc, nu), X
