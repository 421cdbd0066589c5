"""Host-side estimator design (pympc_amd/kalman.py, mirror of pyMPC/kalman.py).  The reference needs the `control`
package (absent here), so the gains are pinned by what defines them: the Riccati equation and the reference's own
consistency check of its __main__ block (kalman.py:157-196)."""
import numpy as np

from pympc_amd.kalman import kalman_design, kalman_design_simple, LinearStateEstimator, BatchLinearStateEstimator


def _point_mass():
    Ts, M, b = 0.2, 2.0, 0.3                     # kalman.py:160-177
    Ad = np.array([[1.0, Ts], [0, 1.0 - b / M * Ts]])
    Bd = np.array([[0.0], [Ts / M]])
    Cd = np.array([[1.0, 0.0]])
    Dd = np.array([[0.0]])
    return Ad, Bd, Cd, Dd


def test_simple_design_solves_the_filter_riccati_equation():
    Ad, Bd, Cd, Dd = _point_mass()
    Q, R = 10 * np.eye(2), np.eye(1)
    L, P, W = kalman_design_simple(Ad, Bd, Cd, Dd, Q, R, type='filter')
    S = Cd @ P @ Cd.T + R
    ric = Ad @ P @ Ad.T - Ad @ P @ Cd.T @ np.linalg.solve(S, Cd @ P @ Ad.T) + Q - P
    assert np.abs(ric).max() < 1e-9 * np.abs(P).max()
    assert np.allclose(L, P @ Cd.T @ np.linalg.inv(S))
    Lp, _, _ = kalman_design_simple(Ad, Bd, Cd, Dd, Q, R, type='predictor')
    assert np.allclose(Lp, Ad @ L)
    assert np.all(np.abs(W) < 1.0)                # estimator poles inside the unit circle
    assert np.all(np.abs(np.linalg.eigvals(Ad - Lp @ Cd)) < 1.0)


def test_general_design_reduces_to_simple_design():
    """kalman.py:186-196: the simple design written in general form gives the same (predictor) gain."""
    Ad, Bd, Cd, Dd = _point_mass()
    Q, R = 10 * np.eye(2), np.eye(1)
    Bk = np.hstack([Bd, np.eye(2)])
    Dk = np.hstack([Dd, np.zeros((1, 2))])
    Lg, Pg, _ = kalman_design(Ad, Bk, Cd, Dk, Q, R)
    Lp, Ps, _ = kalman_design_simple(Ad, Bd, Cd, Dd, Q, R, type='predictor')
    assert np.allclose(Pg, Ps, rtol=1e-9) and np.allclose(Lg, Lp, rtol=1e-9)


def test_estimator_recursion_and_batch_version():
    Ad, Bd, Cd, Dd = _point_mass()
    L, _, _ = kalman_design_simple(Ad, Bd, Cd, Dd, 10 * np.eye(2), np.eye(1))
    rng = np.random.default_rng(0)
    KF = LinearStateEstimator(np.array([0.1, 0.2]), Ad, Bd, Cd, Dd, L)
    KB = BatchLinearStateEstimator(np.array([[0.1, 0.2]] * 3), np.stack([Ad] * 3), np.stack([Bd] * 3), np.stack([Cd] * 3), np.stack([L] * 3))
    x = np.array([0.0, 0.3])
    for _ in range(20):
        u = rng.standard_normal(1)
        y = Cd @ x + 0.01 * rng.standard_normal(1)
        x = Ad @ x + Bd @ u
        KF.update(y); KF.predict(u)
        KB.update(np.stack([y] * 3)); KB.predict(np.stack([u] * 3))
        assert np.allclose(KB.x, KF.x, rtol=1e-13, atol=1e-15)
    assert np.abs(KF.x - x).max() < 0.2             # it tracks
    ys = KF.sim(np.zeros((5, 1)))
    assert ys.shape == (5, 1) and np.allclose(ys[0], Cd @ KF.x)
