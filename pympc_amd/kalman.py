"""State estimation next to the MPC hot path: mirror of ``pyMPC/kalman.py``.

Host side (one-off design, plain numpy/scipy): ``kalman_design`` (kalman.py:25-72), ``kalman_design_simple``
(kalman.py:75-106) -- the reference calls ``control.dare``; the same generalised discrete Riccati equation is solved
here with ``scipy.linalg.solve_discrete_are`` -- and ``LinearStateEstimator`` (kalman.py:109-153: ``predict``,
``update``, ``out_y``, ``sim``).  ``BatchLinearStateEstimator`` holds B of them as stacked arrays; inside the
device-side closed loop (``BatchMPCController.run(..., estimator=...)`` -> ``mpcqp_mpc_loop``) its update/predict
pair runs on the GPU between the plant step and the QP refresh, as in
examples/example_inverted_pendulum_kalman.py:135-174.
"""
import numpy as np
import scipy.linalg as sla


def _dare(A, B, Q, R, S=None):
    """X, closed-loop eigenvalues, gain G = (B'XB+R)^-1 (B'XA+S') of the DARE (the triple ``control.dare`` returns)."""
    X = sla.solve_discrete_are(A, B, Q, R, s=S)
    BtX = B.T @ X
    G = np.linalg.solve(BtX @ B + R, BtX @ A + (S.T if S is not None else 0.0))
    W = np.linalg.eigvals(A - B @ G)
    return X, W, G


def kalman_design(A, B, C, D, Qn, Rn, Nn=None):
    """General Kalman predictor gain for ``x+ = A x + B [u; w]``, ``y = C x + D [u; w] + v`` with ``E[ww'] = Qn``,
    ``E[vv'] = Rn``, ``E[wv'] = Nn`` (same call and return values as kalman.py:25-72: ``(L, P, W)`` = gain, Riccati
    solution, estimator poles).

    Derivation used here: the state sees the noise ``e_x = Bw w`` and the output sees ``e_y = Dw w + v``; the DARE of the
    predictor needs the joint covariance of ``(e_x, e_y)``, which is ``G Qn G'`` for ``G = [Bw; Dw]`` plus the terms
    that involve ``v``."""
    A, B, C, D, Qn, Rn = (np.atleast_2d(np.asarray(M, dtype=float)) for M in (A, B, C, D, Qn, Rn))
    nx, ny, nw = A.shape[0], C.shape[0], Qn.shape[0]
    n_known = B.shape[1] - nw                         # leading columns of B, D belong to the known input u
    G = np.vstack([B[:, n_known:], D[:, n_known:]])   # how w enters (state; output)
    cross = np.zeros((nw, ny)) if Nn is None else np.atleast_2d(np.asarray(Nn, dtype=float))
    J = G @ Qn @ G.T
    Gv = G @ cross                                    # E[(G w) v']
    J[:, nx:] += Gv
    J[nx:, :] += Gv.T
    J[nx:, nx:] += Rn
    J = (J + J.T) / 2
    P, W, K = _dare(A.T, C.T, J[:nx, :nx], J[nx:, nx:], J[:nx, nx:])
    return K.T, P, W


def kalman_design_simple(A, B, C, D, Qn, Rn, type='filter'):
    """Kalman filter / predictor gain for x+ = Ax + Bu + w, y = Cx + Du + v (kalman.py:75-106)."""
    A, C = np.atleast_2d(np.asarray(A, dtype=float)), np.atleast_2d(np.asarray(C, dtype=float))
    Qn, Rn = np.atleast_2d(np.asarray(Qn, dtype=float)), np.atleast_2d(np.asarray(Rn, dtype=float))
    P, W, _ = _dare(A.T, C.T, Qn, Rn)
    if type == 'filter':
        L = P @ C.T @ np.linalg.inv(C @ P @ C.T + Rn)
    elif type == 'predictor':
        L = A @ P @ C.T @ np.linalg.inv(C @ P @ C.T + Rn)
    else:
        raise ValueError("Unknown Kalman design type. Specify either filter or predictor!")
    return L, P, W


class LinearStateEstimator:
    """kalman.py:109-153, same attributes and methods."""

    def __init__(self, x0, A, B, C, D, L):
        self.x = np.copy(x0)
        self.y = C @ self.x
        self.A, self.B, self.C, self.D, self.L = A, B, C, D, L
        self.nx = np.shape(A)[0] if np.size(A) > 1 else 1
        self.nu = np.shape(B)[1] if np.size(B) > 1 else 1
        self.ny = np.shape(C)[0] if np.size(C) > 1 else 1

    def out_y(self, u):
        """Current output estimate (kalman.py:122-123)."""
        return self.y

    def predict(self, u):
        self.x = self.A @ self.x + self.B @ u          # x[k|k] -> x[k+1|k]
        self.y = self.C @ self.x
        return self.x

    def update(self, y_meas):
        self.x = self.x + self.L @ (y_meas - self.y)   # x[k+1|k] -> x[k+1|k+1]
        return self.x

    def sim(self, u_seq, x=None):
        """Open-loop output prediction ``y_i = C x_i + D u_i``, ``x_{i+1} = A x_i + B u_i`` from ``x`` (default: the
        current estimate) over the rows of ``u_seq``; returns ``[len(u_seq), ny]`` (kalman.py:136-152)."""
        from itertools import accumulate
        U = np.asarray(u_seq, dtype=float).reshape(-1, self.nu)
        start = self.x if x is None else x
        states = accumulate(U[:-1], lambda xi, ui: self.A @ xi + self.B @ ui, initial=np.asarray(start, dtype=float))
        return np.array([self.C @ xi + self.D @ ui for xi, ui in zip(states, U)]).reshape(len(U), self.ny)


class BatchLinearStateEstimator:
    """B independent ``LinearStateEstimator``s as stacked arrays: ``A [B,nx,nx]``, ``B [B,nx,nu]``, ``C [B,ny,nx]``,
    ``L [B,nx,ny]``, estimate ``x [B,nx]``; ``x_true [B,nx]`` is the plant state the measurements are taken from when
    the estimator runs inside the device loop, ``v`` an optional measurement-noise sequence ``[nsteps,B,ny]``."""

    def __init__(self, x0, A, B, C, L, x_true=None, v=None):
        self.A, self.Bm, self.C, self.L = (np.ascontiguousarray(M, dtype=float) for M in (A, B, C, L))
        self.x = np.array(x0, dtype=float)
        self.y = np.einsum('bij,bj->bi', self.C, self.x)
        self.x_true = np.ascontiguousarray(self.x.copy() if x_true is None else x_true, dtype=float)
        self.v = v

    def predict(self, u):
        self.x = np.einsum('bij,bj->bi', self.A, self.x) + np.einsum('bij,bj->bi', self.Bm, u)
        self.y = np.einsum('bij,bj->bi', self.C, self.x)
        return self.x

    def update(self, y_meas):
        self.x = self.x + np.einsum('bij,bj->bi', self.L, y_meas - self.y)
        return self.x
