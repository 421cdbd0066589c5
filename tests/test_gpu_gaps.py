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
    1024 cfg-3 instances (what the headline throughput depends on), applied inputs to 1e-6; ALL 512 cfg-5 instances with the differences their
    ill-conditioned KKT systems produce counted and bounded (see the comment at the assertions)."""
    B, nx, nu, Np, xbox, sample, alongside = (1024, 12, 4, 30, 10.0, 128, 1024) if cfg == 'cfg3' else (512, 20, 8, 100, 1.0, 32, 512)
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
    # both batches (cfg-3: the first 20 steps = the driver's timed region; cfg-5: all 50; QP-solves/s = iterations/s / iterations per solve, and the
    # second factor is what this pins).  Worker processes (oracle/cpu_bench.alongside_pool).
    from oracle import cpu_bench
    along = np.arange(B)                                                   # (both batches whole since round 6: 512 cfg-5 instances x 50 steps = 14 s on the box's 16 cores)
    steps = 20 if cfg == 'cfg3' else 50
    res = cpu_bench.alongside_pool(along, tr, 1e-3, nx, nu, Np, xbox, steps=steps)
    assert [r[0] for r in res] == list(along)
    dev_iters, ora_iters = int(tr['iter'][:steps].sum()), sum(r[5] for r in res)
    assert not any(r[3] for r in res), [(r[0], r[3][:3]) for r in res if r[3]][:5]                 # the same status in every solve of every instance
    diffs = [(r[0], k, a, b_) for r in res for k, a, b_ in r[2]]                                      # (instance, step, oracle iterations, device iterations)
    if cfg == 'cfg3':
        # the headline batch: every one of its 20 480 solves has the oracle's iteration count, every applied input the oracle's to 1e-6
        assert not diffs, diffs[:5]
        assert dev_iters == ora_iters
        assert max(r[4] for r in res) <= 1e-6, max(r[4] for r in res)
    else:
        # cfg-5 (tight state box, slack active; OSQP's rho adaptation drives rho to 1e3 .. 8e4 during the cold solve: with rho_eq = 1e3 rho against sigma = 1e-6
        # the KKT condition number reaches 1e13 and a double-precision linear solve is accurate to about the termination tolerance itself): two correct
        # implementations agree on the outcome of every solve, not always on the round in which a residual test passes.  Observed on ALL 512 instances x 50 steps
        # (25 600 solves; round 6): no status differs; 131 instances have at least one count difference, 296 of the 330 differing solves in steps 0 .. 7 (the
        # transient right after the cold start, where rho was just adapted; the largest difference 100 iterations); from step 8 on five instances differ once or
        # twice and ONE (instance 150, rho 3.7e4) sits on the threshold throughout -- the device passes the test at 25 iterations where the oracle needs 50, in 30
        # of its 50 steps; inside bench.py's timed region (steps >= 25) that instance is the only difference: 18 of 12 800 solves.  Totals: device 840 200
        # iterations, oracle 840 725 (-0.06 %).  Of the 312 instances whose rho stays below 1e4, 13 have a difference (the rule "exact wherever rho <= 1e4" that a
        # 33-instance sample suggested does not survive the whole batch).  Applied inputs -- ADMM iterates at tolerance 1e-3, not optima -- beyond
        # max(1e-6, 3e-7 rho) (eps_machine * 1e3 rho / sigma) of the oracle's: 12 instances somewhere, 4 from step 8 on.
        assert all(abs(a - b_) <= 100 for _, _, a, b_ in diffs), [d for d in diffs if abs(d[2] - d[3]) > 100][:5]
        late = {(i, k) for i, k, _, _ in diffs if k >= cpu_bench.LATE_FROM}
        timed = {(i, k) for i, k, _, _ in diffs if k >= 25}
        assert len({i for i, _ in late}) <= 12, sorted({i for i, _ in late})                          # (seen: 6 of 512)
        assert len(timed) <= 0.005 * B * (steps - 25), len(timed)                                    # (seen: 18 of 12 800, all of instance 150)
        assert abs(dev_iters - ora_iters) <= 0.003 * ora_iters, (dev_iters, ora_iters)              # (seen: 840 200 against 840 725)
        bound = lambda r: max(1e-6, 3e-7 * r[1])
        assert sum(r[4] > bound(r) for r in res) <= 30 and sum(r[6] > bound(r) for r in res) <= 10, ([(r[0], r[4]) for r in res if r[4] > bound(r)][:8], [(r[0], r[6]) for r in res if r[6] > bound(r)][:8])
        assert max(r[4] for r in res) <= 0.1, max(r[4] for r in res)                                  # (relative to max(1e-3, |u|): 6.5e-5 absolute on an input near zero)
    print('%s: %d instances alongside for %d steps; %d with rho > 1e4; %d solves with a count difference in %d instances; ADMM iterations device %d / oracle %d'
          % (cfg, len(res), steps, sum(r[1] > 1e4 for r in res), len(diffs), len({d[0] for d in diffs}), dev_iters, ora_iters))


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
