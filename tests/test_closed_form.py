"""The condensed closed-form controller (tests/closed_form.py, after test_scripts/alternative/unconstrained.py:141-183)
as a solver-independent check of the QP path when no inequality is active: on CPU it pins the oracle's SOLVER half
(oracle/osqp_ref.c on the reference-layout sparse QP) against plain dense linear algebra; with -m gpu the HIP path."""
import warnings

import numpy as np
import pytest

from closed_form import unconstrained_mpc, prediction_matrices


def _cases():
    from pympc_amd import fixtures
    inf = np.inf
    out = {}

    def free(kw, wide=None):
        kw = dict(kw)
        nx, nu = kw['Ad'].shape[0], kw['Bd'].shape[1]
        b = inf if wide is None else wide
        kw.update(xmin=-b * np.ones(nx), xmax=b * np.ones(nx), umin=-b * np.ones(nu), umax=b * np.ones(nu),
                  Dumin=-b * np.ones(nu), Dumax=b * np.ones(nu))
        return kw
    out['point_mass_free'] = free(fixtures.point_mass())
    out['point_mass_wide'] = free(fixtures.point_mass(), 1e4)                 # finite bounds, rows present but inactive
    out['quadcopter_free'] = free(fixtures.quadcopter())
    out['random_12_4_30_wide'] = free(fixtures.random_lti(5), 1e3)
    out['random_5_3_8_nc_free'] = free(dict(fixtures.random_lti(12, nx=5, nu=3, Np=8), Nc=3))
    kw = free(fixtures.random_lti(13, nx=4, nu=2, Np=12), 1e3)
    kw['xref'] = 0.3 * np.random.default_rng(1).standard_normal((13, 4))      # time-varying reference
    kw['uref'] = np.array([0.1, -0.2]); kw['uminus1'] = np.array([0.3, 0.1])
    out['random_4_2_12_xref2d'] = kw
    kw = free(dict(fixtures.random_lti(14, nx=20, nu=8, Np=40)), 1e3)         # 32 x 32 stage blocks
    out['random_20_8_40_wide'] = kw
    return out


CASES = _cases()


def test_prediction_matrices_reproduce_a_simulation():
    rng = np.random.default_rng(0)
    Ad, Bd = 0.5 * rng.standard_normal((3, 3)), rng.standard_normal((3, 2))
    x0, U = rng.standard_normal(3), rng.standard_normal((6, 2))
    calA, calB = prediction_matrices(Ad, Bd, 6)
    x, X = x0, []
    for k in range(6):
        x = Ad @ x + Bd @ U[k]; X.append(x)
    assert np.allclose(calA @ x0 + calB @ U.ravel(), np.concatenate(X), rtol=1e-13, atol=1e-13)


def _check(K, kw):
    u_seq, x_seq = unconstrained_mpc(**kw)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        K.setup()
        u0, info = K.output(return_u_seq=True, return_x_seq=True, return_eps_seq=True)
    su, sx = max(1e-3, np.abs(u_seq).max()), max(1e-3, np.abs(x_seq).max())
    assert np.abs(info['u_seq'] - u_seq).max() <= 1e-6 * su
    assert np.abs(info['x_seq'] - x_seq).max() <= 1e-6 * sx
    assert np.abs(info['eps_seq']).max() <= 1e-6 * sx              # no state bound is active: the slack stays at zero
    assert np.abs(u0 - u_seq[0]).max() <= 1e-6 * su
    # one closed-loop step further, warm-started, with the applied input as u_{-1}
    x1 = kw['Ad'] @ np.asarray(kw['x0'], dtype=float) + kw['Bd'] @ u_seq[0]
    kw1 = dict(kw, x0=x1, uminus1=u_seq[0])
    u_seq1, _ = unconstrained_mpc(**kw1)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        K.update(x1, u_seq[0])
    assert np.abs(K.output() - u_seq1[0]).max() <= 1e-6 * max(1e-3, np.abs(u_seq1).max())


@pytest.mark.parametrize('name', sorted(CASES))
def test_oracle_solver_matches_closed_form(name):
    from pympc_amd import MPCController
    from oracle.osqp_oracle import OSQP
    kw = dict(CASES[name], eps_abs=1e-10, eps_rel=1e-10)
    K = MPCController(**kw); K.prob = OSQP(); K.solver_settings = dict(max_iter=400000)
    _check(K, kw)


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(CASES))
def test_gpu_matches_closed_form(name):
    from pympc_amd import MPCController
    kw = dict(CASES[name], eps_abs=1e-10, eps_rel=1e-10)
    K = MPCController(**kw); K.solver_settings = dict(max_iter=400000)
    _check(K, kw)
