"""GPU parity tests: the HIP path (through the C ABI) against
  (a) the structural golden vectors captured from the reference (tests/golden/qp_*.npz),
  (b) the CPU oracle (oracle/osqp_ref.c) on the same inputs, iterate by iterate,
  (c) the certified optimum golden vectors (tests/golden/opt_*.npz) -- the north-star criterion:
      u* within 1e-6 relative of the reference QP's optimum.
Run on the GPU box with:  python -m pytest tests -m gpu
"""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

from util import golden_names, load_golden, golden_kwargs, golden_csc, update_steps, apply_attrs

pytestmark = pytest.mark.gpu

DEVICE_FIXTURES = golden_names()


def _gpu_controller(kw, **settings):
    from pympc_amd import MPCController
    K = apply_attrs(MPCController(**kw), kw)
    K.solver_settings = dict(settings)
    return K


def _oracle_controller(kw, **settings):
    from pympc_amd import MPCController
    from oracle.osqp_oracle import OSQP
    K = apply_attrs(MPCController(**kw), kw)
    K.prob = OSQP()
    K.solver_settings = dict(settings)
    return K


def _eff(P):
    """P as a solver that keeps triu(P) sees it."""
    U = sp.triu(P).toarray()
    return U + np.triu(U, 1).T


def _clip(v):
    return np.clip(v, -1e30, 1e30)


def _rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


@pytest.mark.parametrize('name', DEVICE_FIXTURES)
def test_device_built_qp_matches_reference(name):
    g = load_golden(name)
    K = _gpu_controller(golden_kwargs(g))
    K.setup(solve=False)
    bp = K.prob.batch_problem
    P, q, A, l, u = bp.export_qp()
    assert np.array_equal(P[0], _eff(golden_csc(g, 'P')))
    assert np.array_equal(A[0], golden_csc(g, 'A').toarray())
    assert np.allclose(q[0], g['q'], rtol=2e-15, atol=1e-300)
    assert np.array_equal(l[0], _clip(g['l'])) and np.array_equal(u[0], _clip(g['u']))
    for st in update_steps(g):
        K.update(st['x'], u=st['u'], xref=st['xref'], solve=False)
        _, q, _, l, u = bp.export_qp()
        assert np.allclose(q[0], st['q'], rtol=2e-15, atol=1e-300)
        assert np.array_equal(l[0], _clip(st['l'])) and np.array_equal(u[0], _clip(st['u_bound']))
        if 'upd0_output_u' in g.files and np.array_equal(st['q'], g['upd0_q']):
            # the capture run called output() here with the stub's zero solution (mpc.py:330 side effect)
            K.uminus1_rh = np.array(g['upd0_output_u'])


@pytest.mark.parametrize('name', DEVICE_FIXTURES)
def test_equilibration_matches_oracle(name):
    g = load_golden(name)
    kw = golden_kwargs(g)
    K = _gpu_controller(kw); K.setup(solve=False)
    Ko = _oracle_controller(kw); Ko.setup(solve=False)
    D, E, c, rho = K.prob.batch_problem.scaling()
    Do, Eo, co = Ko.prob.scaling()
    assert _rel(D[0], Do) < 1e-12 and _rel(E[0], Eo) < 1e-12 and abs(c[0] - co) / co < 1e-12
    assert rho[0] == 0.1


@pytest.mark.parametrize('name', DEVICE_FIXTURES)
def test_kkt_solve_matches_dense(name):
    g = load_golden(name)
    K = _gpu_controller(golden_kwargs(g)); K.setup(solve=False)
    bp = K.prob.batch_problem
    D, E, c, rho = bp.scaling()
    P, A = _eff(golden_csc(g, 'P')), golden_csc(g, 'A').toarray()
    l, u = _clip(g['l']), _clip(g['u'])
    ls, us = E[0] * l, E[0] * u
    rho_vec = np.where((ls < -1e26) & (us > 1e26), 1e-6, np.where(us - ls < 1e-4, 1e3 * rho[0], rho[0]))
    Kmat = c[0] * P + np.diag(1e-6 / D[0] ** 2) + A.T @ np.diag(rho_vec * E[0] ** 2) @ A
    rng = np.random.default_rng(5)
    rhs = rng.standard_normal(P.shape[0])
    sol = bp.kkt_solve(rhs[None])[0]
    ref = np.linalg.solve(Kmat, rhs)
    assert _rel(sol, ref) < 1e-8
    assert np.abs(Kmat @ sol - rhs).max() < 1e-8 * max(1.0, np.abs(Kmat).max() * np.abs(sol).max())


@pytest.mark.parametrize('name', DEVICE_FIXTURES)
@pytest.mark.parametrize('iters', [1, 7, 40])
def test_admm_iterates_match_oracle(name, iters):
    g = load_golden(name)
    kw = golden_kwargs(g)
    K = _gpu_controller(kw); K.setup(solve=False)
    Ko = _oracle_controller(kw); Ko.setup(solve=False)
    K.prob.batch_problem.iterate(iters)
    Ko.prob.iterate(iters)
    x, z, y = K.prob.batch_problem.iterate_state()
    xo, zo, yo, _ = Ko.prob.iterate_state()
    assert _rel(x[0], xo) < 1e-8 and _rel(z[0], zo) < 1e-8
    assert np.abs(y[0] - yo).max() < 1e-8 * max(1.0, np.abs(yo).max())


@pytest.mark.parametrize('name', DEVICE_FIXTURES)
def test_default_tolerance_solve_matches_oracle(name):
    """Reference defaults (eps 1e-3, mpc.py:80): same status, same iteration count, same iterate."""
    g = load_golden(name)
    kw = golden_kwargs(g)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K = _gpu_controller(kw); K.setup()
        Ko = _oracle_controller(kw); Ko.setup()
    assert K.res.info.status == Ko.res.info.status
    assert K.res.info.iter == Ko.res.info.iter
    assert K.res.info.rho_updates == Ko.res.info.rho_updates
    assert _rel(K.res.x, Ko.res.x) < 1e-6
    assert abs(K.res.info.obj_val - Ko.res.info.obj_val) <= 1e-7 * max(1.0, abs(Ko.res.info.obj_val))
    assert np.allclose(K.output(), Ko.output(), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('name', DEVICE_FIXTURES)
def test_optimum_parity_1e6(name):
    """North-star criterion: u* within 1e-6 (relative) of the reference QP's certified optimum."""
    g = load_golden(name)
    opt = load_golden(name, prefix='opt_')
    kw = golden_kwargs(g)
    kw.update(eps_abs=1e-11, eps_rel=1e-11)          # the tolerance the goldens were made at (make_optimum.py)
    K = _gpu_controller(kw, max_iter=400000)
    K.setup()
    assert K.res.info.status == 'solved'
    u0 = K.output()
    scale = max(np.abs(opt['u0']).max(), 1e-3)
    assert np.abs(u0 - opt['u0']).max() <= 1e-6 * scale
    assert _rel(K.res.x, opt['x']) < 1e-6
    assert abs(K.res.info.obj_val - float(opt['obj_val'])) <= 1e-8 * max(1.0, abs(float(opt['obj_val'])))


@pytest.mark.parametrize('name', ['point_mass', 'cart_pole', 'quadcopter', 'random_12_4_30'])
def test_closed_loop_matches_oracle(name):
    """Warm-started receding horizon (examples/example_point_mass.py:88-101 pattern), tight tolerance."""
    g = load_golden(name)
    kw = golden_kwargs(g)
    kw.update(eps_abs=1e-9, eps_rel=1e-9)
    K = _gpu_controller(kw, max_iter=100000); K.setup()
    Ko = _oracle_controller(kw, max_iter=100000); Ko.setup()
    Ad, Bd = kw['Ad'], kw['Bd']
    x = np.array(kw['x0'], dtype=float)
    for step in range(12):
        u = K.output()
        uo = Ko.output()
        assert np.abs(u - uo).max() <= 1e-6 * max(np.abs(uo).max(), 1e-3), step
        x = Ad @ x + Bd @ uo
        K.update(x, uo)
        Ko.update(x, uo)
        assert K.res.info.status == 'solved' and Ko.res.info.status == 'solved'


def test_infeasible_problem_reports_like_oracle():
    """u_{-1} far outside the input box with a tight rate limit: the hard rows conflict
    (mpc.py:561-580) -> 'primal infeasible' -> output() falls back to uref (mpc.py:304)."""
    from pympc_amd import fixtures
    kw = fixtures.point_mass()
    kw['uminus1'] = np.array([5.0])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        K = _gpu_controller(kw); K.setup()
        Ko = _oracle_controller(kw); Ko.setup()
    assert Ko.res.info.status == 'primal infeasible'
    assert K.res.info.status == Ko.res.info.status
    assert K.res.info.iter == Ko.res.info.iter
    assert any('OSQP did not solve the problem!' in str(x.message) for x in w)
    assert np.all(np.isnan(K.res.x))
    assert np.array_equal(K.output(), kw['uref'])
    # the solver recovers (cold start) once the problem is feasible again
    K.update(np.array([0.1, 0.2]), np.array([0.0]))
    Ko.update(np.array([0.1, 0.2]), np.array([0.0]))
    assert K.res.info.status == 'solved' and K.res.info.iter == Ko.res.info.iter
    assert np.allclose(K.output(), Ko.output(), rtol=1e-6, atol=1e-9)


def test_batch_matches_per_instance_oracle():
    from pympc_amd import BatchMPCController, fixtures
    B = 6
    kws = [fixtures.random_lti(100 + i) for i in range(B)]
    stack = lambda k: np.stack([kw[k] for kw in kws])
    K = BatchMPCController(stack('Ad'), stack('Bd'), Np=30, x0=stack('x0'), xref=stack('xref'), uref=stack('uref'),
                           uminus1=stack('uminus1'), Qx=stack('Qx'), QxN=stack('QxN'), Qu=stack('Qu'), QDu=stack('QDu'),
                           xmin=stack('xmin'), xmax=stack('xmax'), umin=stack('umin'), umax=stack('umax'),
                           Dumin=stack('Dumin'), Dumax=stack('Dumax'), eps_feas=1e6, eps_abs=1e-9, eps_rel=1e-9)
    K.setup()
    U, info = K.output(return_status=True, return_x_seq=True)
    assert all(s == 'solved' for s in info['status'])
    xs = []
    for i, kw in enumerate(kws):
        kw = dict(kw); kw.update(eps_abs=1e-9, eps_rel=1e-9)
        Ko = _oracle_controller(kw); Ko.setup()
        uo = Ko.output()
        assert np.abs(U[i] - uo).max() <= 1e-6 * max(np.abs(uo).max(), 1e-3)
        xs.append(kw['Ad'] @ kw['x0'] + kw['Bd'] @ uo)
    # one warm-started step for the whole batch
    K.update(np.stack(xs))
    U2 = K.output()
    for i, kw in enumerate(kws):
        kw = dict(kw); kw.update(eps_abs=1e-9, eps_rel=1e-9)
        Ko = _oracle_controller(kw); Ko.setup(); u1 = Ko.output(); Ko.update(xs[i]); uo = Ko.output()
        assert np.abs(U2[i] - uo).max() <= 1e-6 * max(np.abs(uo).max(), 1e-3)


def test_full_size_batch_properties():
    """BASELINE cfg-3 size (1024 x (12,4,30)): size-independent properties of every returned solution:
    exact dynamics consistency, bound satisfaction of the hard rows, and the KKT certificate on a sample."""
    from pympc_amd import BatchMPCController, fixtures
    from util import kkt_certificate
    from pympc_amd.controller import MPCController
    B = 1024
    kws = [fixtures.random_lti(i) for i in range(B)]
    stack = lambda k: np.stack([kw[k] for kw in kws])
    K = BatchMPCController(stack('Ad'), stack('Bd'), Np=30, x0=stack('x0'), Qx=np.eye(12), QxN=np.eye(12),
                           Qu=0.1 * np.eye(4), QDu=0.1 * np.eye(4), xmin=-10 * np.ones(12), xmax=10 * np.ones(12),
                           umin=-np.ones(4), umax=np.ones(4), Dumin=-0.5 * np.ones(4), Dumax=0.5 * np.ones(4),
                           eps_abs=1e-8, eps_rel=1e-8)
    K.setup()
    U, info = K.output(return_status=True, return_x_seq=True, return_u_seq=True)
    assert all(s == 'solved' for s in info['status'])
    X, Us = info['x_seq'], info['u_seq']
    Ad, Bd = stack('Ad'), stack('Bd')
    pred = np.einsum('bij,bkj->bki', Ad, X[:, :-1]) + np.einsum('bij,bkj->bki', Bd, Us)
    assert np.abs(pred - X[:, 1:]).max() < 1e-6
    assert np.abs(X[:, 0] - stack('x0')).max() < 1e-6
    assert Us.max() <= 1 + 1e-6 and Us.min() >= -1 - 1e-6
    x, y, _ = K.prob.solution()
    for i in (0, 511, 1023):
        C = MPCController(**kws[i]); C.prob = object(); C.x0_rh = C.x0; C.uminus1_rh = C.uminus1
        C._compute_QP_matrices_()
        stat, pv, comp = kkt_certificate(C.P, C.q, C.A, C.l, C.u, x[i], y[i])
        assert stat < 1e-6 and pv < 1e-6 and comp < 1e-6


def test_device_resident_inputs_match_host_inputs():
    """bench.py drives the library with torch ROCm tensors (device pointers, caller's stream): same results as
    the numpy (host pointer) path, step by step."""
    import torch
    from pympc_amd import fixtures
    from pympc_amd.solver import BatchProblem
    B, nx, nu, Np = 5, 12, 4, 30
    kws = [fixtures.random_lti(200 + i) for i in range(B)]
    stack = lambda k: np.stack([kw[k] for kw in kws])
    dev = torch.device('cuda', 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    common = dict(eps_abs=1e-6, eps_rel=1e-6)
    ph = BatchProblem(B, nx, nu, Np, **common)
    pd = BatchProblem(B, nx, nu, Np, stream=torch.cuda.current_stream(dev).cuda_stream, **common)
    args = [stack('Ad'), stack('Bd'), stack('Qx'), stack('QxN'), stack('Qu'), stack('QDu'), stack('xmin'), stack('xmax'),
            stack('umin'), stack('umax'), stack('Dumin'), stack('Dumax'), stack('uref'), np.full((B, 1), 1e6),
            stack('x0'), stack('uminus1'), stack('xref')]
    ph.setup(*args)
    pd.setup(*[t(a) for a in args])
    x = stack('x0')
    u_dev = torch.empty((B, nu), dtype=torch.float64, device=dev)
    for step in range(4):
        ph.solve_async(); pd.solve_async()
        uh = ph.u0()
        pd.u0(out=u_dev)
        assert np.array_equal(uh, u_dev.cpu().numpy()), step
        assert [i.iter for i in ph.infos()] == [i.iter for i in pd.infos()]
        x = np.einsum('bij,bj->bi', stack('Ad'), x) + np.einsum('bij,bj->bi', stack('Bd'), uh)
        ph.update(x, uh)
        pd.update(t(x), u_dev)
    it, chk, ref, sol = pd.stats()
    assert sol == 4 * B and it >= 25 * sol and chk >= sol


def _stacked_batch(kws, **kw):
    from pympc_amd import BatchMPCController
    stack = lambda k: np.stack([np.asarray(d[k], dtype=float) for d in kws])
    k0 = kws[0]
    args = dict(Np=k0['Np'], Nc=k0.get('Nc'), x0=stack('x0'), xref=stack('xref'), uref=stack('uref'), uminus1=stack('uminus1'),
                Qx=stack('Qx'), QxN=stack('QxN'), Qu=stack('Qu'), QDu=stack('QDu'), xmin=stack('xmin'), xmax=stack('xmax'),
                umin=stack('umin'), umax=stack('umax'), Dumin=stack('Dumin'), Dumax=stack('Dumax'),
                eps_feas=np.array([[d.get('eps_feas', 1e6)] for d in kws]))
    args.update(kw)
    return BatchMPCController(stack('Ad'), stack('Bd'), **args)


@pytest.mark.parametrize('case', ['random_12_4_30', 'cart_pole', 'quadcopter_nc', 'random_20_8_12'])
def test_device_loop_matches_stepwise_api(case):
    """mpcqp_mpc_run (update -> solve -> output -> plant on the device for K steps) against the same
    closed loop driven step by step through update()/output(): identical inputs, statuses and iteration counts."""
    from pympc_amd import fixtures
    K_STEPS = 12
    if case == 'random_12_4_30':
        kws = [fixtures.random_lti(300 + i) for i in range(7)]
    elif case == 'random_20_8_12':
        kws = [fixtures.random_lti(40 + i, nx=20, nu=8, Np=12, xbox=1.0) for i in range(3)]
    else:
        kws = []
        for i in range(4):
            kw = dict(fixtures.NAMED[case]())
            kw['x0'] = np.asarray(kw['x0'], dtype=float) + 0.01 * i
            kws.append(kw)
    B, nx = len(kws), kws[0]['Ad'].shape[0]
    rng = np.random.default_rng(5)
    w = 0.01 * rng.standard_normal((K_STEPS, B, nx))
    Kd = _stacked_batch(kws); Kd.setup()
    Ks = _stacked_batch(kws); Ks.setup()
    tr = Kd.run(K_STEPS, w=w)
    assert np.array_equal(tr['x'][0], Ks.x0_rh)
    for k in range(K_STEPS):
        u = Ks.output()
        assert np.array_equal(u, tr['u'][k]), k
        xn = np.einsum('bij,bj->bi', Ks.Ad, tr['x'][k]) + np.einsum('bij,bj->bi', Ks.Bd, u) + w[k]
        assert np.allclose(xn, tr['x'][k + 1], rtol=1e-13, atol=1e-14)
        Ks.update(tr['x'][k + 1])                      # the device's own x_{k+1}: no plant rounding differences
        infos = Ks.prob.infos()
        assert [i.status for i in infos] == list(tr['status'][k]), k
        assert [i.iter for i in infos] == list(tr['iter'][k]), k
    # the handle is left exactly where the stepwise controller is
    assert np.array_equal(Kd.output(), Ks.output())
    xs_d, ys_d, _ = Kd.prob.solution(); xs_s, ys_s, _ = Ks.prob.solution()
    assert np.array_equal(xs_d, xs_s) and np.array_equal(ys_d, ys_s)


def test_device_loop_custom_plant_and_failure_fallback():
    """Plant matrices different from the model, and output()'s u_failure branch (mpc.py:271-336) inside the loop:
    with max_iter below the first termination check no solve ends 'solved' -> u = uref."""
    from pympc_amd import fixtures
    kws = [fixtures.random_lti(400 + i) for i in range(3)]
    for kw in kws:
        kw['uref'] = np.array([0.05, -0.02, 0.01, 0.03])
    Ap = np.stack([kw['Ad'] * 0.9 for kw in kws]); Bp = np.stack([kw['Bd'] * 1.1 for kw in kws])
    K = _stacked_batch(kws); K.setup()
    tr = K.run(5, Ap=Ap, Bp=Bp)
    for k in range(5):
        xn = np.einsum('bij,bj->bi', Ap, tr['x'][k]) + np.einsum('bij,bj->bi', Bp, tr['u'][k])
        assert np.allclose(xn, tr['x'][k + 1], rtol=1e-13, atol=1e-14)
    assert (tr['status'] == 1).all()
    Kf = _stacked_batch(kws, max_iter=10); Kf.setup()
    trf = Kf.run(3)
    assert np.isin(trf['status'], (-2, 2)).all() and (trf['iter'] == 10).all()     # max-iter / solved inaccurate: not 'solved'
    assert np.array_equal(trf['u'], np.broadcast_to(np.stack([kw['uref'] for kw in kws]), trf['u'].shape))


def test_device_loop_time_varying_reference():
    """xref_traj inside the device loop == update(x, u, xref_k) step by step (mpc.py:338-364), 1-row and (Np+1)-row xref."""
    from pympc_amd import fixtures
    K_STEPS = 6
    rng = np.random.default_rng(9)
    for rows_full in (False, True):
        kws = [fixtures.random_lti(500 + i) for i in range(3)]
        B, nx, Np = len(kws), 12, 30
        if rows_full:
            for kw in kws:
                kw['xref'] = np.tile(kw['xref'], (Np + 1, 1))
        per = (Np + 1) * nx if rows_full else nx
        xr = 0.1 * rng.standard_normal((K_STEPS, B, per))
        Kd = _stacked_batch(kws); Kd.setup()
        Ks = _stacked_batch(kws); Ks.setup()
        tr = Kd.run(K_STEPS, xref_traj=xr)
        for k in range(K_STEPS):
            u = Ks.output()
            assert np.array_equal(u, tr['u'][k]), (rows_full, k)
            Ks.update(tr['x'][k + 1], xref=xr[k].reshape((B, Np + 1, nx) if rows_full else (B, nx)))
            assert [i.iter for i in Ks.prob.infos()] == list(tr['iter'][k])
        assert np.array_equal(Kd.output(), Ks.output())


def test_device_loop_output_feedback_matches_reference_style_loop():
    """Output feedback inside the device loop (LinearStateEstimator, pyMPC/kalman.py:109-134) against the loop of
    examples/example_inverted_pendulum_kalman.py:135-174 driven from the host: numpy estimator + stepwise controller."""
    from pympc_amd import fixtures
    from pympc_amd.kalman import kalman_design_simple, BatchLinearStateEstimator
    K_STEPS = 10
    kws = []
    for i in range(3):
        kw = dict(fixtures.cart_pole())
        kw['x0'] = np.asarray(kw['x0'], dtype=float) * (1.0 + 0.1 * i)
        kws.append(kw)
    B, nx, nu = len(kws), 4, 1
    Ad, Bd = kws[0]['Ad'], kws[0]['Bd']
    Cd = np.array([[1.0, 0, 0, 0], [0, 0, 1.0, 0]])            # position and angle are measured
    L, _, _ = kalman_design_simple(Ad, Bd, Cd, np.zeros((2, 1)), np.diag([0.1, 10, 0.1, 10]), 0.01 * np.eye(2), type='filter')
    rng = np.random.default_rng(3)
    v = 0.01 * rng.standard_normal((K_STEPS, B, 2))
    w = 0.001 * rng.standard_normal((K_STEPS, B, nx))
    x_true0 = np.stack([kw['x0'] * 1.05 for kw in kws])         # the controller starts from a wrong estimate
    st = lambda M: np.stack([M] * B)
    est = BatchLinearStateEstimator(np.stack([kw['x0'] for kw in kws]), st(Ad), st(Bd), st(Cd), st(L), x_true=x_true0.copy(), v=v)
    Kd = _stacked_batch(kws); Kd.setup()
    tr = Kd.run(K_STEPS, w=w, estimator=est)
    # host replica: numpy estimator, stepwise controller
    Ks = _stacked_batch(kws); Ks.setup()
    KF = BatchLinearStateEstimator(np.stack([kw['x0'] for kw in kws]), st(Ad), st(Bd), st(Cd), st(L))
    x = x_true0.copy()
    for k in range(K_STEPS):
        y = np.einsum('bij,bj->bi', st(Cd), x) + v[k]
        u = Ks.output()
        assert np.array_equal(u, tr['u'][k]), k
        assert np.allclose(x, tr['x'][k], rtol=1e-12, atol=1e-14) and np.allclose(y, tr['y'][k], rtol=1e-12, atol=1e-14)
        assert np.allclose(KF.x, tr['xhat'][k], rtol=1e-12, atol=1e-14)
        x = np.einsum('bij,bj->bi', st(Ad), x) + np.einsum('bij,bj->bi', st(Bd), u) + w[k]
        KF.update(y); KF.predict(u)
        assert np.allclose(KF.x, tr['xhat'][k + 1], rtol=1e-11, atol=1e-13)
        x = tr['x'][k + 1]; KF.x = tr['xhat'][k + 1].copy(); KF.y = np.einsum('bij,bj->bi', st(Cd), KF.x)   # no rounding drift
        Ks.update(tr['xhat'][k + 1])
    assert np.array_equal(est.x_true, tr['x'][-1]) and np.array_equal(Kd.x0_rh, tr['xhat'][-1])


@pytest.mark.parametrize('dims', [(1, 1, 2, 2), (2, 1, 3, 1), (3, 2, 5, 5), (13, 3, 6, 6), (12, 5, 4, 2), (24, 8, 4, 4), (20, 12, 3, 3),
                                  (4, 1, 150, 75), (2, 2, 64, 64), (3, 12, 20, 20), (3, 12, 22, 22)])
def test_boundary_dimensions_match_oracle(dims):
    """Shapes at the edges of the device code paths: smallest problem (nx=nu=1, Np=2), nx+nu = 16/17 (block size switch),
    nx+nu = 32 (largest 32-wide stage), Nc = 1, long horizons with Nc < Np (the reference's Kalman example uses Np=150, Nc=75),
    LDS-resident and global-memory iterate, and the two sides of the owner map's limit (one input element per thread:
    (3,12,20) has 252 input elements and rows, (3,12,22) 276 and falls back to the global-memory passes although it would fit
    LDS).  u* against the oracle at tight tolerance, plus one warm step."""
    from pympc_amd import fixtures
    nx, nu, Np, Nc = dims
    kw = dict(fixtures.random_lti(900 + nx * 7 + nu, nx=nx, nu=nu, Np=Np, xbox=3.0))
    kw['x0'] = 0.3 * kw['x0']
    if Nc != Np:
        kw['Nc'] = Nc
    kw.update(eps_abs=1e-9, eps_rel=1e-9)
    K = _gpu_controller(kw, max_iter=200000); Ko = _oracle_controller(kw, max_iter=200000)
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


def test_unsupported_dimensions_fail_loudly():
    from pympc_amd import fixtures
    kw = dict(fixtures.random_lti(1, nx=120, nu=12, Np=3))      # nx + nu > 128 (32 < nx + nu <= 128: tests/test_gpu_wide.py)
    K = _gpu_controller(kw)
    with pytest.raises(NotImplementedError):
        K.setup()


def test_device_loop_recovers_from_infeasible_step_like_stepwise():
    """An instance whose first QP is primal infeasible (u_{-1} far outside what the input box and the Delta-u bounds allow):
    'primal infeasible' -> output() falls back to u_failure = uref (mpc.py:271-336), the iterate is cold-started, and the
    next step is feasible again -- identical inside the device loop and through the stepwise API."""
    from pympc_amd import fixtures
    kws = [fixtures.random_lti(600 + i) for i in range(4)]
    kws[1]['uminus1'] = np.array([5.0, 0.0, 0.0, 0.0])        # needs u_0[0] in [4.5, 5.5] but u <= 1
    kws[3]['uminus1'] = np.array([0.0, -7.0, 0.0, 0.0])
    Kd = _stacked_batch(kws); Ks = _stacked_batch(kws)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Kd.setup(); Ks.setup()
        st0 = Ks.status()
        assert st0[1] == 'primal infeasible' and st0[3] == 'primal infeasible' and st0[0] == 'solved' and st0[2] == 'solved'
        tr = Kd.run(6)
        for k in range(6):
            u = Ks.output()
            assert np.array_equal(u, tr['u'][k]), k
            if k == 0:
                assert np.array_equal(u[1], kws[1]['uref']) and np.array_equal(u[3], kws[3]['uref'])
            Ks.update(tr['x'][k + 1])
            infos = Ks.prob.infos()
            assert [i.status for i in infos] == list(tr['status'][k]) and [i.iter for i in infos] == list(tr['iter'][k])
        assert (tr['status'][1:] == 1).all()


def test_device_loop_long_run_stays_bounded_and_solved():
    """400 noisy closed-loop steps of 64 instances in four launches: every solve ends 'solved', states stay inside a
    generous box, and chaining launches equals one long launch."""
    from pympc_amd import fixtures
    kws = [fixtures.random_lti(700 + i) for i in range(64)]
    rng = np.random.default_rng(11)
    w = 0.05 * rng.standard_normal((400, 64, 12))
    Ka = _stacked_batch(kws); Ka.setup()
    Kb = _stacked_batch(kws); Kb.setup()
    ta = Ka.run(400, w=w)
    parts = [Kb.run(100, w=w[100 * i:100 * (i + 1)]) for i in range(4)]
    assert np.array_equal(ta['u'], np.concatenate([p['u'] for p in parts]))
    assert np.array_equal(ta['x'][-1], parts[-1]['x'][-1])
    assert (ta['status'] == 1).all()
    assert np.isfinite(ta['x']).all() and np.abs(ta['x']).max() < 50 and np.abs(ta['u']).max() <= 1 + 1e-6


def test_mpc_step_equals_update_then_output():
    """mpcqp_mpc_step (u = K(x, u_{-1}), mpc.py:377-384) against update() + output(), including the implicit u_{-1} hand-over
    and the u_failure fallback."""
    from pympc_amd import fixtures
    kws = [fixtures.random_lti(800 + i) for i in range(5)]
    Ka = _stacked_batch(kws); Ka.setup()
    Kb = _stacked_batch(kws); Kb.setup()
    x = Ka.x0_rh.copy()
    rng = np.random.default_rng(2)
    ua, ub = Ka.output(), Kb.output()
    assert np.array_equal(ua, ub)
    for k in range(5):
        x = np.einsum('bij,bj->bi', Ka.Ad, x) + np.einsum('bij,bj->bi', Ka.Bd, ua) + 0.01 * rng.standard_normal(x.shape)
        Ka.update(x); ua = Ka.output()
        ub = Kb.step(x, ub if k == 0 else None)            # u_{-1} handed over by the previous step from k = 1 on
        assert np.array_equal(ua, ub), k
    Kf = _stacked_batch(kws, max_iter=10); Kf.setup()
    uf = Kf.step(x)
    assert np.array_equal(uf, Kf.uref)                     # not 'solved' after 10 iterations -> u_failure


def test_load_balancing_map_never_changes_results(monkeypatch):
    """The workgroup -> instance map (rebuilt from the observed iteration counts) is pure scheduling: a batch larger than
    the number of CUs gives bit-identical trajectories, statuses and iteration counts with and without it, through the
    device loop and through the stepwise API."""
    from pympc_amd import fixtures
    B = 300                                              # > 256 CUs: the map becomes a real permutation
    kws = [fixtures.random_lti(1000 + i) for i in range(B)]
    rng = np.random.default_rng(4)
    w = 0.02 * rng.standard_normal((30, B, 12))
    out = {}
    for flag in ('1', '0'):
        from pympc_amd import _lib
        from pympc_amd.solver import forced_settings
        with forced_settings(tuning=0 if flag == '1' else _lib.TUNE_NO_BALANCE):
            K = _stacked_batch(kws); K.setup()
        parts = [K.run(10, w=w[10 * i:10 * (i + 1)]) for i in range(3)]       # the map is rebuilt after every launch
        x = parts[-1]['x'][-1]
        us = []
        for k in range(20):                              # stepwise: rebuilt every 16 solves
            u = K.output(); us.append(u)
            x = np.einsum('bij,bj->bi', K.Ad, x) + np.einsum('bij,bj->bi', K.Bd, u)
            K.update(x)
        out[flag] = (np.concatenate([p['u'] for p in parts]), np.concatenate([p['iter'] for p in parts]), np.array(us), K.prob.solution()[0])
    for a, b in zip(out['1'], out['0']):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('B,steps', [(300, 12), (1100, 7)])
def test_persistent_queue_never_changes_results(B, steps):
    """Beyond one workgroup per resident slot a closed-loop launch is PERSISTENT: as many workgroups as slots, each taking (instance, step range)
    items off a queue (k_mpc_run, mpcqp_kernels.h).  Pure scheduling: the trajectories, statuses and iteration counts of two consecutive launches are
    bit-identical with the queue and its step-range parts (default), with whole closed loops as items (TUNE_NO_PARTS), and with the hardware's own
    dispatch of one workgroup per instance (TUNE_NO_QUEUE).  300 instances: more than the 256 one-at-a-time slots of the register-resident kernel;
    1100 with the bandwidth kernel forced: more than its 1024."""
    from pympc_amd import fixtures, _lib
    from pympc_amd.solver import forced_settings
    kws = [fixtures.random_lti(2000 + i) for i in range(B)]
    rng = np.random.default_rng(8)
    w = 0.02 * rng.standard_normal((2 * steps, B, 12))
    out = []
    for tuning in (0, _lib.TUNE_NO_PARTS, _lib.TUNE_NO_QUEUE):
        with forced_settings(tuning=tuning, **({'backend': 'sweeps'} if B > 1024 else {})):
            K = _stacked_batch(kws); K.setup()
        parts = [K.run(steps, w=w[steps * i:steps * (i + 1)]) for i in range(2)]
        out.append(tuple(np.concatenate([p[k] for p in parts]) for k in ('u', 'x', 'iter', 'status')) + (K.prob.solution()[0],))
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert np.array_equal(a, b)


def test_three_quarter_slots_never_change_results():
    """Round 6: closed-loop launches of the 32 x 32 bandwidth kernel with its iterate in memory (BASELINE configs[4]'s kernel) that would hold (nearly) every
    instance resident at once run on three quarters of their slots with the (instance, step range) queue instead (run_grid, mpcqp.hip: the stragglers start
    first and run faster).  400 instances of a (20, 8, 40) controller on a 256-unit part: 384 slots instead of 512.  Bit-identical to the hardware's dispatch of
    one workgroup per instance (TUNE_NO_QUEUE) and to a forced slot count (the development switch, 10 eighths per unit)."""
    from pympc_amd import fixtures, _lib
    from pympc_amd.solver import forced_settings
    B, steps = 400, 8
    kws = [fixtures.random_lti(4000 + i, nx=20, nu=8, Np=40, xbox=2.0) for i in range(B)]
    rng = np.random.default_rng(9)
    w = 0.02 * rng.standard_normal((2 * steps, B, 20))
    out = []
    for tuning in (0, _lib.TUNE_NO_QUEUE, 10 << 24):
        with forced_settings(tuning=tuning):
            K = _stacked_batch(kws); K.setup()
        assert K.prob.kernel_name(True).replace(' ', '').startswith('k_mpc_run<32,false'), K.prob.kernel_name(True)
        parts = [K.run(steps, w=w[steps * i:steps * (i + 1)]) for i in range(2)]
        out.append(tuple(np.concatenate([p[k] for p in parts]) for k in ('u', 'x', 'iter', 'status')) + (K.prob.solution()[0],))
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert np.array_equal(a, b)


def _random_case(seed):
    """A seeded random controller: dimensions, horizon split, bound pattern (finite / one-sided / absent), weights
    (including semidefinite ones) and reference shape are all drawn."""
    from pympc_amd import fixtures
    rng = np.random.default_rng(7000 + seed)
    nx = int(rng.integers(1, 14)); nu = int(rng.integers(1, 6))
    Np = int(rng.integers(2, 26)); Nc = int(rng.integers(1, Np + 1)) if rng.random() < 0.5 else Np
    kw = dict(fixtures.random_lti(9000 + seed, nx=nx, nu=nu, Np=Np, xbox=4.0))
    kw['x0'] = 0.5 * kw['x0']
    if Nc != Np:
        kw['Nc'] = Nc
    inf = np.inf
    def pattern(lo, hi):
        lo, hi = lo.copy(), hi.copy()
        for i in range(lo.size):
            t = rng.random()
            if t < 0.2: lo[i] = -inf
            elif t < 0.4: hi[i] = inf
            elif t < 0.5: lo[i], hi[i] = -inf, inf
        return lo, hi
    kw['xmin'], kw['xmax'] = pattern(kw['xmin'], kw['xmax'])
    kw['Dumin'], kw['Dumax'] = pattern(kw['Dumin'], kw['Dumax'])
    if rng.random() < 0.3:
        kw['umin'], kw['umax'] = pattern(kw['umin'], kw['umax'])
    w = rng.random()
    if w < 0.25: kw['QDu'] = np.zeros((nu, nu))                       # Qu > 0 keeps the optimum unique
    elif w < 0.5: kw['Qu'] = np.zeros((nu, nu))                       # QDu > 0 does
    G = rng.standard_normal((nx, nx)); kw['Qx'] = G @ G.T / nx        # dense PSD state weight
    if rng.random() < 0.5: kw['QxN'] = 3.0 * kw['Qx']
    if rng.random() < 0.3: kw['xref'] = 0.2 * rng.standard_normal((Np + 1, nx))
    else: kw['xref'] = 0.2 * rng.standard_normal(nx)
    kw['uref'] = 0.1 * rng.standard_normal(nu)
    kw['uminus1'] = 0.1 * rng.standard_normal(nu)
    kw['eps_feas'] = float(10.0 ** rng.integers(2, 7))
    return kw


@pytest.mark.parametrize('seed', range(24))
def test_randomised_controllers_match_oracle(seed):
    """Seeded random controllers (shapes, Nc < Np, one-sided and absent bounds, semidefinite weights, 1-D / 2-D references):
    cold solve and one warm step against the oracle at tight tolerance; statuses must agree too."""
    kw = _random_case(seed)
    kw.update(eps_abs=1e-10, eps_rel=1e-10)     # the reference author's own parity setting (test_scripts/main_du.py:125)
    K = _gpu_controller(kw, max_iter=400000); Ko = _oracle_controller(kw, max_iter=400000)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup(); Ko.setup()
        assert K.res.info.status == Ko.res.info.status
        (u, info), (uo, infoo) = K.output(return_u_seq=True), Ko.output(return_u_seq=True)
        scale = max(1e-3, np.abs(infoo['u_seq']).max())
        assert np.abs(info['u_seq'] - infoo['u_seq']).max() <= 1e-6 * scale
        x = kw['Ad'] @ kw['x0'] + kw['Bd'] @ uo
        K.update(x, uo); Ko.update(x, uo)
        assert K.res.info.status == Ko.res.info.status
        assert np.abs(K.output() - Ko.output()).max() <= 1e-6 * scale


def test_step_after_output_uses_the_output_as_previous_input():
    """setup() -> output() -> step(x): update(x, u=None) of the reference uses the input returned by the last output()
    as u_{-1} (mpc.py:330,357-359).  The one-call step must do the same although output() only changed the HOST copy."""
    from pympc_amd import fixtures
    kws = [fixtures.random_lti(820 + i) for i in range(4)]
    for kw in kws:
        kw['uminus1'] = np.array([0.4, -0.3, 0.2, 0.1])       # differs from what output() will return
    Ka = _stacked_batch(kws); Ka.setup()
    Kb = _stacked_batch(kws); Kb.setup()
    ua, ub = Ka.output(), Kb.output()
    x = np.einsum('bij,bj->bi', Ka.Ad, Ka.x0_rh) + np.einsum('bij,bj->bi', Ka.Bd, ua)
    Ka.update(x); ua = Ka.output()
    ub = Kb.step(x)                                            # no u given: must pick up the output() above
    assert np.array_equal(ua, ub)
    x = np.einsum('bij,bj->bi', Ka.Ad, x) + np.einsum('bij,bj->bi', Ka.Bd, ua)
    Ka.update(x); ua = Ka.output()
    assert np.array_equal(ua, Kb.step(x))                      # ... and from then on the device's own copy


def test_loop_reference_shape_is_checked_and_may_change():
    """xref_traj must hold nx or (Np+1)*nx values per step; a loop may switch the reference shape like update(x, u, xref)
    may (mpc.py:414-424), and then equals the stepwise calls."""
    from pympc_amd import fixtures
    kws = [fixtures.random_lti(830 + i) for i in range(2)]
    B, nx, Np = 2, 12, 30
    Kd = _stacked_batch(kws); Kd.setup()
    Ks = _stacked_batch(kws); Ks.setup()
    rng = np.random.default_rng(6)
    with pytest.raises(ValueError):
        Kd.run(3, xref_traj=np.zeros((3, B, 5)))
    xr = 0.1 * rng.standard_normal((4, B, Np + 1, nx))         # constant reference at setup, time-varying in the loop
    tr = Kd.run(4, xref_traj=xr)
    for k in range(4):
        u = Ks.output()
        assert np.array_equal(u, tr['u'][k]), k
        Ks.update(tr['x'][k + 1], xref=xr[k])
    assert np.array_equal(Kd.output(), Ks.output())
    from pympc_amd import _lib
    import ctypes as C
    io = _lib.Loop(); z = np.zeros((1, B, nx)); io.xref_traj = z.ctypes.data; io.xref_rows = 7
    assert Kd.prob._L.mpcqp_mpc_loop(Kd.prob._h, 1, C.byref(io)) == -1        # MPCQP_ERR_ARG


def test_warm_start_sets_z_like_osqp():
    """osqp.warm_start(x=, y=) also sets z = A x; the next solve must then iterate exactly like the oracle's."""
    from pympc_amd import fixtures
    kw = dict(fixtures.random_lti(840))
    K = _gpu_controller(kw); K.setup()
    Ko = _oracle_controller(kw); Ko.setup()
    rng = np.random.default_rng(8)
    x, y = 0.1 * rng.standard_normal(K.prob.n), 0.1 * rng.standard_normal(K.prob.m)
    K.prob.warm_start(x=x, y=y); Ko.prob.warm_start(x=x, y=y)
    K.prob.batch_problem.iterate(3); Ko.prob.iterate(3)
    xg, zg, yg = K.prob.batch_problem.iterate_state()
    xo, zo, yo, _ = Ko.prob.iterate_state()
    assert _rel(xg[0], xo) < 1e-8 and _rel(zg[0], zo) < 1e-8 and np.abs(yg[0] - yo).max() < 1e-8 * max(1.0, np.abs(yo).max())


@pytest.mark.parametrize('name', ['quadcopter', 'random_20_8_12', 'point_mass_nc'])
def test_refactor_rewrites_the_same_factor(name):
    """mpcqp_refactor (what bench.py times as the cost of one rho update) recomputes the factor in place from the current rho:
    the reduced-KKT solve and the next solve are bit-identical with and without it."""
    g = load_golden(name)
    kw = golden_kwargs(g)
    K = _gpu_controller(kw); K.setup(solve=True)
    bp = K.prob.batch_problem
    rhs = np.random.default_rng(11).standard_normal((1, bp.n))
    before = bp.kkt_solve(rhs)
    bp.refactor(); bp.synchronize()
    assert np.array_equal(bp.kkt_solve(rhs), before)
    K2 = _gpu_controller(kw); K2.setup(solve=True)
    x = np.asarray(kw['x0'], float)
    for _ in range(3):
        u = K.output(); u2 = K2.output()
        assert np.array_equal(u, u2)
        x = kw['Ad'] @ x + kw['Bd'] @ u
        K.prob.batch_problem.refactor()
        K.update(x, u); K2.update(x, u)


def test_persistent_queue_with_output_feedback_and_a_moving_reference():
    """The same invariance for the closed loop's optional inputs: 300 cart-pole controllers (more than the 256 one-at-a-time slots of their dense kernel) with a
    LinearStateEstimator in the loop, measurement and process noise and a reference that changes every step -- an instance's estimate, true state, previous input and
    reference cross from one queue item to the next through memory exactly as they cross from one launch to the next."""
    from pympc_amd import fixtures, _lib
    from pympc_amd.kalman import BatchLinearStateEstimator
    from pympc_amd.solver import forced_settings
    B, steps = 300, 9
    kw0 = fixtures.cart_pole()
    kw0.setdefault('uref', np.zeros(1)); kw0.setdefault('uminus1', kw0['uref']); kw0.setdefault('QxN', kw0['Qx'])
    rng = np.random.default_rng(21)
    kws = [dict(kw0, x0=kw0['x0'] + 0.02 * rng.standard_normal(4)) for _ in range(B)]
    C = np.tile(np.array([[1.0, 0, 0, 0], [0, 0, 1.0, 0]]), (B, 1, 1))
    Lg = np.tile(0.3 * np.array([[1.0, 0], [0.5, 0], [0, 1.0], [0, 0.5]]), (B, 1, 1))
    st = lambda k: np.stack([np.asarray(d[k], dtype=float) for d in kws])
    w, v = 1e-3 * rng.standard_normal((steps, B, 4)), 1e-3 * rng.standard_normal((steps, B, 2))
    xref = np.tile(np.asarray(kw0['xref'], dtype=float), (steps, B, 1)); xref[:, :, 0] += 0.01 * np.arange(steps)[:, None]
    out = []
    for tuning in (0, _lib.TUNE_NO_QUEUE):
        with forced_settings(tuning=tuning):
            K = _stacked_batch(kws); K.setup()
        est = BatchLinearStateEstimator(st('x0'), st('Ad'), st('Bd'), C, Lg, x_true=st('x0').copy(), v=v)
        tr = K.run(steps, w=w, xref_traj=xref, estimator=est)
        out.append((tr['u'], tr['x'], tr['xhat'], tr['y'], tr['iter'], tr['status'], est.x_true.copy()))
    for a, b in zip(*out):
        assert np.array_equal(a, b)

