"""Driver-visible tests for what round 2 only printed in bench output or tested in a weakened form:
  * the BENCH workloads themselves -- cfg-3 (1024 x (12,4,30)) and the UNMODIFIED cfg-5 batch (512 x (20,8,100), instance 276
    included) -- run as bench.py runs them (eps 1e-3, 50 warm closed-loop steps in the device loop), then sampled instances'
    u* at the parity tolerance against the oracle: the north-star criterion (1e-6 relative) on the benchmarked batches;
  * the swapped eps kwargs of mpc.py:266 (eps_abs != eps_rel) through the drop-in class on the GPU, against the oracle;
  * seam B (the caller's P, q, A, l, u and update(q, l, u) per step) inside a closed loop, against the reference-class trajectory."""
import warnings

import numpy as np
import pytest

from util import load_traj

pytestmark = pytest.mark.gpu


def _bench_batch(B, nx, nu, Np, xbox, eps):
    from pympc_amd import BatchMPCController, fixtures
    kws = [fixtures.random_lti(i, nx=nx, nu=nu, Np=Np, xbox=xbox) for i in range(B)]
    stack = lambda k: np.stack([np.asarray(kw[k], dtype=float) for kw in kws])
    K = BatchMPCController(stack('Ad'), stack('Bd'), Np=Np, x0=stack('x0'), Qx=np.eye(nx), QxN=np.eye(nx), Qu=0.1 * np.eye(nu), QDu=0.1 * np.eye(nu),
                           xmin=-xbox * np.ones(nx), xmax=xbox * np.ones(nx), umin=-np.ones(nu), umax=np.ones(nu),
                           Dumin=-0.5 * np.ones(nu), Dumax=0.5 * np.ones(nu), eps_feas=1e6, eps_abs=eps, eps_rel=eps)
    return K, kws


def _oracle_u(kw, x0, um1):
    from pympc_amd import MPCController
    from oracle.osqp_oracle import OSQP
    kw = dict(kw); kw.update(x0=x0, uminus1=um1, eps_abs=1e-10, eps_rel=1e-10)
    K = MPCController(**kw); K.prob = OSQP(); K.solver_settings = dict(max_iter=400000)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup()
    return K.output(), K.res.info.status


@pytest.mark.parametrize('cfg', ['cfg3', 'cfg5'])
def test_benchmarked_batches_meet_the_north_star_tolerance(cfg):
    """The batches bench.py times, run as bench.py runs them (reference tolerance 1e-3, 50 warm closed-loop steps inside the device
    loop), then (i) at the parity tolerance the u* of 128 of the 1024 cfg-3 instances / 33 of the 512 cfg-5 instances against the
    oracle at 1e-10 (north-star criterion: 1e-6 relative), and (ii) the warm steps themselves at the default tolerance against
    the oracle stepping alongside on the device's own states: same status and the same ADMM iteration count in EVERY step of ALL
    1024 cfg-3 instances (what the headline throughput depends on; 33 of the 512 at cfg-5), applied inputs to 1e-6."""
    B, nx, nu, Np, xbox, sample, alongside = (1024, 12, 4, 30, 10.0, 128, 1024) if cfg == 'cfg3' else (512, 20, 8, 100, 1.0, 32, 32)
    K, kws = _bench_batch(B, nx, nu, Np, xbox, 1e-3)
    rng = np.random.default_rng(11)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup()
        tr = K.run(50, w=0.01 * rng.standard_normal((50, B, nx)))            # the bench's regime: reference tolerance, warm receding horizon
        assert (tr['status'][-25:] == 1).all(), np.argwhere(tr['status'][-25:] != 1)[:5]
        assert tr['iter'].mean() < (45 if cfg == 'cfg3' else 35)
        x, um1 = tr['x'][-1], tr['u'][-1]
        K.prob.update_settings(eps_abs=1e-9, eps_rel=1e-9, max_iter=200000)
        K.update(x, um1)
        U, st = K.output(return_status=True)
    idx = np.unique(np.linspace(0, B - 1, sample).astype(int))
    if cfg == 'cfg5':
        idx = np.unique(np.append(idx, 276))                               # the instance round 2's test replaced
    worst = 0.0
    for i in idx:
        uo, so = _oracle_u(kws[int(i)], x[i], um1[i])
        assert so == 'solved' and st['status'][int(i)] == 'solved', (i, so, st['status'][int(i)])
        worst = max(worst, np.abs(U[i] - uo).max() / max(1e-3, np.abs(uo).max()))
    assert worst <= 1e-6, worst                                            # north_star: u* within 1e-6 relative of the reference solver's
    # (ii) the warm steps at the default tolerance: the oracle on the same states, warm-starting from its own previous iterate -- on EVERY instance of
    # the headline batch (cfg-3: all 1024, the first 20 steps = the driver's timed region; QP-solves/s = iterations/s / iterations per solve, and the
    # second factor is what this pins), on a spread of 32 of the 512 cfg-5 instances (all 50 steps).  Worker processes (oracle/cpu_bench.alongside_pool).
    # Where OSQP's rho adaptation has driven rho beyond 1e4 during the cold solve (cfg-5, tight state box: rho_eq = 1e3 rho against sigma = 1e-6 puts the
    # KKT condition number beyond 1e13), a double-precision linear solve is only accurate to about the termination tolerance itself: two correct
    # implementations then agree on the outcome, not on the round in which a residual test passes.  Rule: EVERY instance with rho <= 1e4 has the oracle's
    # status and iteration count in every step; the instances above are listed with their observed count differences (at most three rounds, in the first
    # steps after the cold start only); everybody's applied inputs -- ADMM iterates at tolerance 1e-3, not optima -- to the accuracy the KKT solve itself
    # has at that rho (eps_machine * 1e3 rho / sigma ~ 3e-7 rho; 1e-6 at best).
    from oracle import cpu_bench
    along = np.arange(B) if cfg == 'cfg3' else np.unique(np.append(np.linspace(0, B - 1, alongside).astype(int), 276))
    steps = 20 if cfg == 'cfg3' else 50
    res = cpu_bench.alongside_pool(along, tr, 1e-3, nx, nu, Np, xbox, steps=steps)
    assert [r[0] for r in res] == list(along)
    high, dev_iters, ora_iters = [], 0, 0
    for i, rho, bad_it, bad_st, worst, total in res:
        assert not bad_st, (cfg, i, rho, bad_st[:3])
        dev_iters += int(tr['iter'][:steps, i].sum()); ora_iters += total
        assert worst <= max(1e-6, 3e-7 * rho), (cfg, i, rho, worst)
        if rho <= 1e4:
            assert not bad_it, (cfg, i, rho, bad_it[:3])
        else:
            # observed (round 6, 33 instances, 11 of them above 1e4): differences of 25 .. 75 iterations in steps 0 .. 5 only -- the transient right after the
            # cold start, where rho was just adapted -- and none from step 6 on (bench.py's timed region starts at step 25)
            high.append((i, rho, [(k, a - b_) for k, a, b_ in bad_it]))
            assert all(k < 8 and abs(a - b_) <= 75 for k, a, b_ in bad_it), (cfg, i, rho, bad_it[:6])
    print('%s: %d instances alongside for %d steps; %d with rho > 1e4: %s; ADMM iterations device %d / oracle %d'
          % (cfg, len(res), steps, len(high), [(i, '%.3g' % r, d) for i, r, d in high][:12], dev_iters, ora_iters))
    if cfg == 'cfg3':
        assert not high, high[:5]                       # (the headline batch never gets there: every one of its 20 480 solves has the oracle's count)
        assert dev_iters == ora_iters
    else:
        assert len(high) <= len(res) // 2, (len(high), len(res))
        assert abs(dev_iters - ora_iters) <= 0.01 * ora_iters, (dev_iters, ora_iters)      # (seen: 58 275 against 57 950)


@pytest.mark.parametrize('name', ['cart_pole', 'accel_brake', 'quadcopter'])
def test_swapped_tolerance_kwargs_reach_the_device(name):
    """mpc.py:266 passes eps_abs=self.eps_rel, eps_rel=self.eps_abs.  With eps_abs=1e-7, eps_rel=1e-3 the solver therefore runs with
    (abs 1e-3, rel 1e-7): the GPU must terminate exactly where the oracle does with THOSE settings -- on these fixtures the
    unswapped reading needs a different number of iterations (checked here too, so the test can tell the two apart)."""
    from pympc_amd import MPCController, fixtures
    from oracle.osqp_oracle import OSQP
    kw = dict(fixtures.NAMED[name]()); kw.update(eps_abs=1e-7, eps_rel=1e-3)
    K = MPCController(**kw)
    Ko = MPCController(**kw); Ko.prob = OSQP()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup(); Ko.setup()
        assert (K.res.info.status, K.res.info.iter) == (Ko.res.info.status, Ko.res.info.iter)
        assert np.abs(K.res.x - Ko.res.x).max() <= 1e-6 * max(1.0, np.abs(Ko.res.x).max())
        O = OSQP(); O.setup(Ko.P, Ko.q, Ko.A, Ko.l, Ko.u, eps_abs=1e-7, eps_rel=1e-3)      # the UNswapped reading
        ru = O.solve()
    assert ru.info.iter != Ko.res.info.iter


class _VectorsOnly:
    """What pyMPC's own class sees of its solver (mpc.py:266,454,369): setup(P, q, A, l, u, **settings), update(q=, l=, u=), solve()."""

    def __init__(self):
        from pympc_amd.solver import DeviceProblem
        self._d = DeviceProblem()

    def setup(self, P, q, A, l, u, mpc=None, **settings):
        self._d.setup(P, q, A, l, u, **settings)

    def update(self, q=None, l=None, u=None, mpc_step=None):
        self._d.update(q=q, l=l, u=u)

    def solve(self):
        return self._d.solve()


@pytest.mark.parametrize('name', ['point_mass', 'accel_brake'])
def test_seam_b_closed_loop_follows_the_reference_class(name):
    """The one-line patch of INTEGRATION.md B (self.prob = DeviceProblem()) in a closed loop: host-built P, q, A, l, u go to the
    library (mpcqp_create_csc / mpcqp_setup_csc), every step's q, l, u through mpcqp_update_vectors; the applied inputs follow
    the trajectory the REFERENCE class produced (tests/golden/traj_*.npz) to 1e-6."""
    from pympc_amd import MPCController, fixtures
    g = load_traj(name)
    kw = dict(fixtures.NAMED[str(g['fixture'])]())
    kw.update(eps_abs=1e-9, eps_rel=1e-9)
    K = MPCController(**kw); K.prob = _VectorsOnly(); K.solver_settings = dict(max_iter=400000)
    xs, us = g['x'], g['u']
    pattern = str(g['pattern'])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup()
        x, u = np.array(kw['x0'], dtype=float), np.array(kw['uminus1'], dtype=float)
        for k in range(min(20, len(us))):
            if pattern == 'update_output':
                K.update(x, u); u = K.output()
            else:
                u = K.output()
            assert np.abs(u - us[k]).max() <= 1e-6 * max(1e-3, np.abs(us).max()), (k, u, us[k])
            x = xs[k + 1]
            if pattern != 'update_output':
                K.update(x)


@pytest.mark.parametrize('B', [128, 512])
def test_auto_selected_latency_backend_at_per_gpu_batches_matches_oracle(B):
    """BASELINE configs[3] read literally leaves 128 instances per GPU (1024 / 8); up to two instances per compute unit the library
    chooses a cyclic-reduction backend on its own (mpcqp_create).  That regime against the oracle: cold solve at the parity tolerance
    (u* of 32 sampled instances to 1e-6 relative), then 10 warm closed-loop steps inside the device loop at the default tolerance
    with the oracle stepping alongside (status and iteration count of every step, inputs to 1e-7)."""
    from pympc_amd import MPCController
    from oracle.osqp_oracle import OSQP
    nx, nu, Np = 12, 4, 30
    K, kws = _bench_batch(B, nx, nu, Np, 10.0, 1e-9)
    K.solver_settings = dict(max_iter=200000)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup()
    # chosen by the library, not forced: 31 stages of cyclic reduction on 512-thread workgroups with a dense top (MODE_BCRT, mpcqp_w8.hip) -- up to
    # three instances per compute unit
    assert K.prob.kernel_name(loop=True) == 'w8::k_mpc_run<16,true,12,4,231,true>', K.prob.kernel_name(loop=True)
    U, st = K.output(return_status=True)
    idx = np.unique(np.linspace(0, B - 1, 32).astype(int))
    worst = 0.0
    for i in idx:
        uo, so = _oracle_u(kws[int(i)], kws[int(i)]['x0'], np.zeros(nu))
        assert so == 'solved' and st['status'][int(i)] == 'solved', (i, so, st['status'][int(i)])
        worst = max(worst, np.abs(U[i] - uo).max() / max(1e-3, np.abs(uo).max()))
    assert worst <= 1e-6, worst
    K2, _ = _bench_batch(B, nx, nu, Np, 10.0, 1e-3)
    rng = np.random.default_rng(5)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K2.setup()
        tr = K2.run(10, w=0.01 * rng.standard_normal((10, B, nx)))
    for i in idx[::4]:
        kw = dict(kws[int(i)]); kw.update(eps_abs=1e-3, eps_rel=1e-3)
        Ko = MPCController(**kw); Ko.prob = OSQP()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            Ko.setup()
            for k in range(10):
                uo = Ko.output()
                assert np.abs(tr['u'][k, i] - uo).max() <= 1e-7 * max(1e-3, np.abs(uo).max()), (B, i, k)
                Ko.update(tr['x'][k + 1, i], tr['u'][k, i])
                assert (Ko.res.info.iter, Ko.res.info.status_val) == (tr['iter'][k, i], tr['status'][k, i]), (B, i, k)


@pytest.mark.parametrize('rho', [0.1, 3.8e4])
@pytest.mark.parametrize('name', ['random_20_8_100', 'random_12_4_30', 'cart_pole_kalman'])
def test_kkt_solve_backward_error_also_at_the_rho_a_tight_state_box_drives_it_to(name, rho):
    """cfg-5's tight state box drives OSQP's rho to ~ 4e4 (rho_eq / sigma = 4e13): there a double-precision KKT solve is only as accurate in the
    FORWARD sense as the tolerance itself, which is why the closed-loop comparison above allows those instances' iteration counts to differ by
    a round.  What must hold at ANY rho is the backward error of the device's solve -- the residual of K sol = rhs on the reference-built
    matrices, relative to |K| |sol| + |rhs| -- so a factorization or refactorization defect at high rho cannot hide behind that allowance."""
    import scipy.sparse as sp
    from pympc_amd import MPCController
    from util import load_golden, golden_kwargs, golden_csc, apply_attrs
    g = load_golden(name)
    K = apply_attrs(MPCController(**golden_kwargs(g)), golden_kwargs(g))
    K.solver_settings = dict(rho=rho, adaptive_rho=0)
    K.setup(solve=False)
    bp = K.prob.batch_problem
    D, E, c, rho_dev = bp.scaling()
    assert abs(rho_dev[0] - rho) <= 1e-12 * rho
    U = sp.triu(golden_csc(g, 'P')).tocsc(); P = U + sp.triu(U, 1).T
    A = golden_csc(g, 'A').tocsc()
    l, u = np.clip(g['l'], -1e30, 1e30), np.clip(g['u'], -1e30, 1e30)
    ls, us = E[0] * l, E[0] * u
    rho_vec = np.where((ls < -1e26) & (us > 1e26), 1e-6, np.where(us - ls < 1e-4, 1e3 * rho, rho))
    Kmat = (c[0] * P + sp.diags(1e-6 / D[0] ** 2) + A.T @ sp.diags(rho_vec * E[0] ** 2) @ A).tocsr()
    rng = np.random.default_rng(9)
    for _ in range(3):
        rhs = rng.standard_normal(P.shape[0])
        sol = bp.kkt_solve(rhs[None])[0]
        assert np.isfinite(sol).all()
        back = np.abs(Kmat @ sol - rhs).max() / (abs(Kmat).dot(np.abs(sol)).max() + np.abs(rhs).max())
        assert back <= 1e-12, (name, rho, back)
    bp.refactor(); bp.synchronize()                                       # the factorization run from inside k_mpc_run must do as well
    sol = bp.kkt_solve(rhs[None])[0]
    assert np.abs(Kmat @ sol - rhs).max() / (abs(Kmat).dot(np.abs(sol)).max() + np.abs(rhs).max()) <= 1e-12
