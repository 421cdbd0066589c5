"""pympc_amd.unconstrained: the gains of the MPC law without inequality constraints (test_scripts/alternative/unconstrained.py:170-183),
computed on the device by multiplier sweeps with the KKT backend (mpcqp_eq_solve), against the condensed closed form restated in tests/closed_form.py (dense normal
equations, shares nothing with the solver) -- and against the constrained controller itself where no constraint is active."""
import warnings

import numpy as np
import pytest

from closed_form import unconstrained_mpc

pytestmark = pytest.mark.gpu


def _closed_form_gains(kw):
    nx, nu = np.asarray(kw['Bd']).shape
    Np, Nc = kw['Np'], kw.get('Nc') or kw['Np']
    cols = []
    for which, size in (('x0', nx), ('xref', nx), ('uref', nu), ('uminus1', nu)):
        for j in range(size):
            args = dict(x0=np.zeros(nx), xref=np.zeros(nx), uref=np.zeros(nu), uminus1=np.zeros(nu))
            args[which] = np.eye(size)[j]
            u_seq, _ = unconstrained_mpc(kw['Ad'], kw['Bd'], Np, Nc=Nc, Qx=kw['Qx'], QxN=kw.get('QxN'), Qu=kw['Qu'], QDu=kw['QDu'], **args)
            cols.append(u_seq.ravel())
    U = np.array(cols).T
    return dict(K_x0=U[:, :nx], K_xref=U[:, nx:2 * nx], K_uref=U[:, 2 * nx:2 * nx + nu], K_um1=U[:, 2 * nx + nu:])


@pytest.mark.parametrize('case', ['cart_pole', 'quadcopter', 'random_12_4_30', 'point_mass_nc', 'random_20_8_12', 'wide_40_8_10'])
def test_device_gains_equal_the_condensed_closed_form(case):
    from pympc_amd import MPCController, fixtures
    if case == 'random_12_4_30': kw = fixtures.random_lti(3)
    elif case == 'random_20_8_12': kw = fixtures.random_lti(5, nx=20, nu=8, Np=12)
    elif case == 'wide_40_8_10': kw = fixtures.random_lti(7, nx=40, nu=8, Np=10)
    elif case == 'point_mass_nc': kw = fixtures.point_mass_nc()
    else: kw = getattr(fixtures, case)()
    K = MPCController(**kw)
    G = K.unconstrained_gains()
    R = _closed_form_gains(kw)
    for name in ('K_x0', 'K_xref', 'K_uref', 'K_um1'):
        assert G[name].shape == R[name].shape
        assert np.abs(G[name] - R[name]).max() <= 1e-9 * max(1.0, np.abs(R[name]).max()), name
    assert K._gain_solver.sweeps <= 20, K._gain_solver.sweeps             # one factorization, a handful of KKT solves -- not an ADMM run
    G2 = K.unconstrained_gains()                                           # the kept handle: same answer, no new allocation
    assert K._gain_solver.prob is not None and all(np.array_equal(G[k], G2[k]) for k in G)


def test_gains_reproduce_the_controller_where_no_constraint_is_active():
    """Small state, wide bounds: the constrained controller's optimum is the linear law."""
    from pympc_amd import MPCController, fixtures
    kw = dict(fixtures.random_lti(11)); kw['x0'] = 0.01 * kw['x0']
    kw.update(eps_abs=1e-10, eps_rel=1e-10)
    K = MPCController(**kw); K.solver_settings = dict(max_iter=400000)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup()
        u, info = K.output(return_u_seq=True)
    G = K.unconstrained_gains()
    U = G['K_x0'] @ kw['x0'] + G['K_xref'] @ kw['xref'] + G['K_uref'] @ kw['uref'] + G['K_um1'] @ kw['uminus1']
    assert np.abs(U - info['u_seq'].ravel()).max() <= 1e-7 * max(1e-3, np.abs(U).max())
