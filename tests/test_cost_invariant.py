"""The cost re-derivation check of the reference's test_scripts/verify_MPC.py:113-145, for the formulation pyMPC/mpc.py builds today: from the predicted sequences
output() returns, recompute every term of the MPC cost by hand -- state, terminal state, input, input increments (with the held last input for Nc < Np), slack --
walk the dynamics along them, and compare with what the solver reports: obj_val (= solver objective + J_CNST, mpc.py:327, 411-440) must be that sum.
(J_CNST weighs the reference with QxN in EVERY stage -- mpc.py:426 -- so the identity is exact for Qx = QxN, which is what every example of the reference uses.)"""
import warnings

import numpy as np
import pytest

NAMES = ['point_mass', 'cart_pole', 'quadcopter', 'random_12_4_30', 'point_mass_nc', 'quadcopter_nc', 'random_5_3_8']


def _check(K, kw):
    nx, nu = np.asarray(kw['Ad']).shape[0], np.asarray(kw['Bd']).reshape(np.asarray(kw['Ad']).shape[0], -1).shape[1]
    Np, Nc = kw['Np'], kw.get('Nc') or kw['Np']
    K.COMPUTE_J_CNST = True
    K.solver_settings = dict(max_iter=400000)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup()
        u0, info = K.output(return_x_seq=True, return_u_seq=True, return_eps_seq=True, return_obj_val=True)
    X, U, E = info['x_seq'], info['u_seq'], info['eps_seq']
    assert X.shape == (Np + 1, nx) and U.shape == (Nc, nu) and E.shape == (Np + 1, nx)
    Ad, Bd = np.asarray(kw['Ad'], dtype=float), np.asarray(kw['Bd'], dtype=float).reshape(nx, nu)
    xref, uref, um1 = (np.asarray(kw[k], dtype=float) for k in ('xref', 'uref', 'uminus1'))
    Xr = xref if xref.ndim == 2 else np.broadcast_to(xref, (Np + 1, nx))
    Qx, QxN, Qu, QDu = (np.asarray(kw[k], dtype=float) for k in ('Qx', 'QxN', 'Qu', 'QDu'))
    uk = lambda k: U[min(k, Nc - 1)]
    J = 0.0
    x = np.asarray(kw['x0'], dtype=float)
    scale = max(1.0, np.abs(X).max())
    for k in range(Np):
        assert np.abs(X[k] - x).max() <= 1e-7 * scale, k                        # the predicted states ARE the model's (equality rows)
        J += 0.5 * (X[k] - Xr[k]) @ Qx @ (X[k] - Xr[k]) + 0.5 * (uk(k) - uref) @ Qu @ (uk(k) - uref)
        x = Ad @ X[k] + Bd @ uk(k)
    assert np.abs(X[Np] - x).max() <= 1e-7 * scale
    J += 0.5 * (X[Np] - Xr[Np]) @ QxN @ (X[Np] - Xr[Np])
    prev = um1
    for k in range(Nc):
        J += 0.5 * (U[k] - prev) @ QDu @ (U[k] - prev)
        prev = U[k]
    J += 0.5 * float(kw.get('eps_feas', 1e6)) * float((E ** 2).sum())
    assert np.array_equal(u0, U[0])
    assert abs(info['obj_val'] - J) <= 1e-6 * max(1.0, abs(J)), (info['obj_val'], J)


def _kw(name):
    from pympc_amd import fixtures
    kw, attrs = fixtures.split_attrs(dict(fixtures.NAMED[name](), eps_abs=1e-10, eps_rel=1e-10))
    assert not attrs and np.array_equal(np.asarray(kw['Qx']), np.asarray(kw['QxN']))
    return kw


@pytest.mark.parametrize('name', NAMES)
def test_reported_cost_is_the_cost_of_the_reported_sequences_oracle(name):
    from pympc_amd import MPCController
    from oracle.osqp_oracle import OSQP
    kw = _kw(name)
    K = MPCController(**kw); K.prob = OSQP()
    _check(K, kw)


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_reported_cost_is_the_cost_of_the_reported_sequences_device(name):
    from pympc_amd import MPCController
    kw = _kw(name)
    _check(MPCController(**kw), kw)
