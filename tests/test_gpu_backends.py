"""The KKT system of an ADMM iteration has more than one solver on the device -- the block-tridiagonal sweeps (every size),
the dense register-resident inverse (N (nx+nu) <= 128: the reference's own examples), block cyclic reduction with a register-resident
factor (16 x 16 stages, up to 31 of them) -- and mpcqp_create picks one by problem size and batch.  Everything must hold for EVERY backend a problem is eligible for: the same tests as tests/test_gpu_parity.py, with
mpcqp_settings.backend forcing the choice (pympc_amd.solver.forced_settings), plus bit-level agreement of the paths that must not depend on it."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

from util import golden_names, load_golden, golden_kwargs, golden_csc, apply_attrs

pytestmark = pytest.mark.gpu


def backend(**forced):
    """Every controller built inside the block gets these mpcqp_settings fields (the KKT backend, include/mpcqp.h: enum mpcqp_backend)."""
    from pympc_amd.solver import forced_settings
    return forced_settings(**forced)


def _dense_eligible(name):
    g = load_golden(name)
    kw = golden_kwargs(g)
    nx, nu = np.atleast_2d(kw['Bd']).shape if np.ndim(kw['Bd']) == 2 else (np.asarray(kw['Ad']).shape[0], 1)
    Np = kw['Np']
    return (Np + 1) * (nx + nu) <= 128 and kw.get('Nc', Np) in (None, Np)


def _bcr_schedule(name):
    """Stage count of the cyclic-reduction schedule mpcqp_create gives the fixture (11, 21 or 31 >= Np + 1), or 0: 16 x 16 stages,
    no held input, at most 31 stages."""
    kw = golden_kwargs(load_golden(name))
    nx, nu = np.atleast_2d(kw['Bd']).shape if np.ndim(kw['Bd']) == 2 else (np.asarray(kw['Ad']).shape[0], 1)
    Np = kw['Np']
    if nx + nu > 16 or kw.get('Nc', Np) not in (None, Np) or Np + 1 > 31:
        return 0
    return 11 if Np + 1 <= 11 else 21 if Np + 1 <= 21 else 31


SMALL = [n for n in golden_names() if _dense_eligible(n)]
SWEEPS, DENSE, BCR, BCR8 = dict(backend='sweeps'), dict(backend='dense'), dict(backend='bcr'), dict(backend='bcr8')
BACKENDS = [SWEEPS, DENSE]
IDS = ['sweeps', 'dense']
# block cyclic reduction (register-resident factor): any fixture with 16 x 16 stages, Nc = Np and at most 31 stages -- the BASELINE shape (12, 4, 30) with
# compile-time dimensions, everything else (the reference's examples, the quadcopter, soft and hard state boxes) through the generic instantiations
BCR_NAMES = [n for n in golden_names() if _bcr_schedule(n)]
# ... and, for the same fixtures, the dense-top factor (mpcqp_latw.h) on 512-thread workgroups ('bcr8': what mpcqp_create picks for at most one
# instance per compute unit) and on 256-thread ones ('bcrt')
CASES = ([(n, e, i) for n in SMALL for e, i in zip(BACKENDS, IDS)] + [(n, SWEEPS, 'sweeps') for n in BCR_NAMES if n not in SMALL] + [(n, BCR, 'bcr') for n in BCR_NAMES]
         + [(n, BCR8, 'bcr8') for n in BCR_NAMES] + [(n, dict(backend='bcrt'), 'bcrt') for n in BCR_NAMES])
CASE_IDS = ['%s-%s' % (n, i) for n, e, i in CASES]


def _ctrl(kw, oracle=False, **settings):
    from pympc_amd import MPCController
    K = apply_attrs(MPCController(**kw), kw)
    if oracle:
        from oracle.osqp_oracle import OSQP
        K.prob = OSQP()
    K.solver_settings = dict(settings)
    return K


def _rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


def test_some_fixture_is_small_enough():
    assert len(SMALL) >= 2, SMALL
    assert {_bcr_schedule(n) for n in BCR_NAMES} == {11, 21, 31}, BCR_NAMES          # every schedule is exercised


@pytest.mark.parametrize('name,env,tag', CASES, ids=CASE_IDS)
def test_backend_is_the_one_asked_for(name, env, tag):
    with backend(**env):
        K = _ctrl(golden_kwargs(load_golden(name))); K.setup(solve=False)
        kn = K.prob.batch_problem.kernel_name(loop=False)
    assert kn.split(',')[4] == {'sweeps': '0', 'dense': '2', 'bcr': str(100 + _bcr_schedule(name)), 'bcr8': str(200 + _bcr_schedule(name)),
                                'bcrt': str(200 + _bcr_schedule(name))}[tag], kn
    assert kn.startswith('w8::') == (tag == 'bcr8'), kn                     # the 512-thread kernels live in the second translation unit


@pytest.mark.parametrize('name,env,tag', CASES, ids=CASE_IDS)
def test_kkt_solve_matches_dense_numpy(name, env, tag):
    g = load_golden(name)
    with backend(**env):
        K = _ctrl(golden_kwargs(g)); K.setup(solve=False)
        bp = K.prob.batch_problem
        D, E, c, rho = bp.scaling()
        U = sp.triu(golden_csc(g, 'P')).toarray(); P = U + np.triu(U, 1).T
        A = golden_csc(g, 'A').toarray()
        l, u = np.clip(g['l'], -1e30, 1e30), np.clip(g['u'], -1e30, 1e30)
        ls, us = E[0] * l, E[0] * u
        rho_vec = np.where((ls < -1e26) & (us > 1e26), 1e-6, np.where(us - ls < 1e-4, 1e3 * rho[0], rho[0]))
        Kmat = c[0] * P + np.diag(1e-6 / D[0] ** 2) + A.T @ np.diag(rho_vec * E[0] ** 2) @ A
        rng = np.random.default_rng(5)
        for _ in range(3):
            rhs = rng.standard_normal(P.shape[0])
            sol = bp.kkt_solve(rhs[None])[0]
            assert _rel(sol, np.linalg.solve(Kmat, rhs)) < 1e-8


@pytest.mark.parametrize('iters', [1, 7, 40])
@pytest.mark.parametrize('name,env,tag', CASES, ids=CASE_IDS)
def test_admm_iterates_match_oracle(name, env, tag, iters):
    kw = golden_kwargs(load_golden(name))
    with backend(**env):
        K = _ctrl(kw); K.setup(solve=False)
        K.prob.batch_problem.iterate(iters)
        x, z, y = K.prob.batch_problem.iterate_state()
    Ko = _ctrl(kw, oracle=True); Ko.setup(solve=False)
    Ko.prob.iterate(iters)
    xo, zo, yo, _ = Ko.prob.iterate_state()
    assert _rel(x[0], xo) < 1e-8 and _rel(z[0], zo) < 1e-8
    assert np.abs(y[0] - yo).max() < 1e-8 * max(1.0, np.abs(yo).max())


@pytest.mark.parametrize('name,env,tag', CASES, ids=CASE_IDS)
def test_default_tolerance_solve_and_optimum(name, env, tag):
    """eps 1e-3 (mpc.py:80): status, iteration count, rho updates as the oracle; eps 1e-9: u* within 1e-6 of the certified optimum."""
    g, opt = load_golden(name), load_golden(name, prefix='opt_')
    kw = golden_kwargs(g)
    with backend(**env), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K = _ctrl(kw); K.setup()
        Ko = _ctrl(kw, oracle=True); Ko.setup()
        assert K.res.info.status == Ko.res.info.status and K.res.info.iter == Ko.res.info.iter
        assert K.res.info.rho_updates == Ko.res.info.rho_updates
        kw2 = type(kw)(kw); kw2.attrs = kw.attrs
        kw2.update(eps_abs=1e-9, eps_rel=1e-9)
        K = _ctrl(kw2, max_iter=200000); K.setup()
        assert K.res.info.status == 'solved'
        assert np.abs(K.output() - opt['u0']).max() <= 1e-6 * max(1e-3, np.abs(opt['u0']).max())      # north-star tolerance


@pytest.mark.parametrize('name', ['point_mass', 'cart_pole'])
def test_closed_loop_is_the_same_on_both_backends(name):
    """40 closed-loop steps stepwise and inside the device loop: the dense backend follows the sweeps to 1e-9 (two different
    linear-algebra routes through the same ADMM iteration) and reproduces the reference-class trajectory to 1e-6."""
    from pympc_amd import BatchMPCController, fixtures
    kw = getattr(fixtures, name)()
    kw.update(eps_abs=1e-9, eps_rel=1e-9)
    out = {}
    for env, tag in zip(BACKENDS, IDS):
        with backend(**env), warnings.catch_warnings():
            warnings.simplefilter('ignore')
            st = lambda a: np.asarray(a, dtype=float)[None]
            Kb = BatchMPCController(st(kw['Ad']), st(kw['Bd']), Np=kw['Np'], x0=st(kw['x0']), xref=st(kw['xref']), uref=st(kw['uref']),
                                    uminus1=st(kw['uminus1']), Qx=st(kw['Qx']), QxN=st(kw['QxN']), Qu=st(kw['Qu']), QDu=st(kw['QDu']),
                                    xmin=st(kw['xmin']), xmax=st(kw['xmax']), umin=st(kw['umin']), umax=st(kw['umax']),
                                    Dumin=st(kw['Dumin']), Dumax=st(kw['Dumax']), eps_feas=kw.get('eps_feas', 1e6),
                                    eps_abs=1e-9, eps_rel=1e-9, max_iter=200000)
            Kb.setup()
            out[tag] = Kb.run(40)
    a, b = out['sweeps'], out['dense']
    assert np.array_equal(a['status'], b['status'])
    assert np.abs(a['u'] - b['u']).max() <= 1e-7 * max(1.0, np.abs(a['u']).max())
    assert np.abs(a['x'] - b['x']).max() <= 1e-7 * max(1.0, np.abs(a['x']).max())


def test_held_multi_input_on_the_dense_backend_is_reproducible():
    """Nc < Np with nu > 1 on the dense backend (random (4,2,10), Nc = 3, four controllers): the assembly of the held input's rows and
    columns used to store two differently-ordered sums to the same entry of K from two threads, so the factor -- and every iterate
    after it -- depended on which store landed last (1e-13 from handle to handle; found by scripts/fuzz_loop.py in round 4).  Two
    handles must now agree bit for bit, and the device loop with the stepwise API as everywhere else."""
    from pympc_amd import BatchMPCController, fixtures
    nx, nu, Np, Nc, B = 4, 2, 10, 3, 4
    kws = [fixtures.random_lti(53007 + i, nx=nx, nu=nu, Np=Np, xbox=4.0) for i in range(B)]
    stack = lambda k: np.stack([np.asarray(kw[k], dtype=float) for kw in kws])
    keys = ('x0', 'xref', 'uref', 'uminus1', 'Qx', 'QxN', 'Qu', 'QDu', 'xmin', 'xmax', 'umin', 'umax', 'Dumin', 'Dumax')

    def make():
        K = BatchMPCController(stack('Ad'), stack('Bd'), Np=Np, Nc=Nc, **{k: stack(k) for k in keys})
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            K.setup()
        return K
    Ka, Kb, Kc = make(), make(), make()
    assert Ka.prob.kernel_name(loop=False).split(',')[4] == '2'               # the dense backend
    xa, xb = Ka.prob.solution()[0], Kb.prob.solution()[0]
    assert np.array_equal(xa, xb) and np.array_equal(xa, Kc.prob.solution()[0])
    w = 0.01 * np.random.default_rng(3).standard_normal((6, B, nx))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        tr = Ka.run(6, w=w)
        for k in range(6):
            assert np.array_equal(Kb.output(), tr['u'][k]), k
            Kb.update(tr['x'][k + 1])
            assert [i.iter for i in Kb.prob.infos()] == list(tr['iter'][k])


def test_a_backend_the_shape_is_not_eligible_for_is_refused():
    """mpcqp_settings.backend forces; it never silently falls back: the dense inverse for 864 unknowns, cyclic reduction for a 41-stage
    horizon or a held input -> MPCQP_ERR_UNSUPPORTED (NotImplementedError through the Python layer)."""
    from pympc_amd import MPCController, fixtures
    for forced, kw in (('dense', fixtures.random_lti(3)), ('bcr', fixtures.random_lti(3, Np=40)), ('bcr8', dict(fixtures.random_lti(3, nx=5, nu=3, Np=8), Nc=4))):
        with backend(backend=forced):
            with pytest.raises(NotImplementedError):
                MPCController(**kw).setup(solve=False)


def test_auto_follows_the_measured_cross_overs():
    """mpcqp_create with backend AUTO: the dense register-resident inverse (N (nx+nu) <= 128) up to six instances per compute unit; block cyclic
    reduction on 512-thread workgroups at EVERY batch size for horizons of 21..30 steps with stages wider than 8 (its five-barrier form against the
    bandwidth kernel: ahead on all six shapes measured), up to three instances per compute unit for anything else it is eligible for; the bandwidth
    kernel beyond (mpcqp.hip, LAB_NOTES.md).  Only handles are created: nothing is solved."""
    import torch
    from pympc_amd.solver import BatchProblem
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    mode = lambda kn: int(kn.split(',')[4])
    for (nx, nu, Np) in ((12, 4, 30), (12, 4, 21), (6, 3, 28)):
        for B in (1, 3 * ncu + 1, 4096):
            kn = BatchProblem(B, nx, nu, Np).kernel_name(True)
            assert kn.startswith('w8::') and mode(kn) == 231, (nx, nu, Np, B, kn)
    for (nx, nu, Np) in ((12, 4, 20), (6, 2, 20), (6, 2, 28), (12, 4, 10)):
        for B, latency in ((3 * ncu, True), (3 * ncu + 1, False)):
            kn = BatchProblem(B, nx, nu, Np).kernel_name(True)
            assert kn.startswith('w8::') == latency and (mode(kn) >= 200) == latency, (nx, nu, Np, B, kn)
    for (nx, nu, Np) in ((4, 1, 20), (3, 1, 30)):
        for B, dense in ((6 * ncu, True), (6 * ncu + 1, False)):
            kn = BatchProblem(B, nx, nu, Np).kernel_name(True)
            assert (mode(kn) == 2) == dense and not kn.startswith('w8::'), (nx, nu, Np, B, kn)


@pytest.mark.parametrize('tag', ['bcr', 'bcr8', 'bcrt'])
def test_cyclic_reduction_device_loop_is_the_stepwise_api(tag):
    """A batch of the BASELINE shape on each cyclic-reduction kernel: 12 closed-loop steps inside the device loop equal the same steps through
    output() / update() bit for bit (inputs, statuses, iteration counts), and the inputs agree with the auto-selected backend's to 1e-9 at a
    tight tolerance (different elimination orders of the same KKT solve)."""
    from pympc_amd import fixtures
    from test_gpu_parity import _stacked_batch
    B = 24
    kws = [fixtures.random_lti(1000 + i) for i in range(B)]
    w = 0.01 * np.stack([fixtures.random_lti_noise_rng(1000 + i).standard_normal((12, 12)) for i in range(B)], axis=1)      # [step][instance][nx]
    tight = dict(eps_abs=1e-9, eps_rel=1e-9, max_iter=20000)
    out = {}
    for t in (tag, 'sweeps'):
        with backend(backend=t), warnings.catch_warnings():
            warnings.simplefilter('ignore')
            K = _stacked_batch(kws, **tight); K.setup()
            tr = K.run(12, w=w)
            K2 = _stacked_batch(kws, **tight); K2.setup()
            for k in range(12):
                assert np.array_equal(K2.output(), tr['u'][k]), (t, k)
                K2.update(tr['x'][k + 1])
                assert [i.iter for i in K2.prob.infos()] == list(tr['iter'][k]), (t, k)
            out[t] = tr
    assert (out[tag]['status'] == 1).all()
    assert np.abs(out[tag]['u'] - out['sweeps']['u']).max() <= 1e-7 * max(1.0, np.abs(out['sweeps']['u']).max())


@pytest.mark.parametrize('tag', ['bcr', 'bcr8', 'bcrt'])
@pytest.mark.parametrize('name', ['random_12_4_30', 'quadcopter', 'cart_pole', 'random_12_4_30_hard'])
def test_refactorization_inside_the_cyclic_reduction_kernels_reproduces_the_setup_factor(name, tag):
    """The factorization run from inside k_mpc_run (mpcqp_refactor: what a rho update does) -- at 512 threads in mpcqp_w8.hip, with the dense top
    inverted in the LDS of the solve kernel -- must give the factor k_setup gave at 256 threads: the KKT solve before and after agrees to rounding
    (every schedule: 31, 11 and 21 stages; soft and hard state box)."""
    g = load_golden(name)
    with backend(backend=tag), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K = _ctrl(golden_kwargs(g)); K.setup(solve=False)
        bp = K.prob.batch_problem
        rhs = np.random.default_rng(2).standard_normal((1, bp.n))
        s0 = bp.kkt_solve(rhs)
        bp.refactor(); bp.synchronize()
        s1 = bp.kkt_solve(rhs)
        assert np.isfinite(s1).all()
        assert np.abs(s0 - s1).max() <= 1e-12 * np.abs(s0).max()
        K2 = _ctrl(golden_kwargs(g)); K2.setup()              # ... and a cold solve with its rho updates ends 'solved' with finite numbers
        assert K2.res.info.status == 'solved' and np.isfinite(K2.res.x).all()


def test_launch_times_bracket_every_step_of_the_device_loop():
    """mpcqp_get_launch_times: per instance, entry <= end of step 0 <= ... <= end of step K-1 <= exit on one clock (100 MHz), exits within the
    HIP-event duration of the launch -- what bench.py's tail split (roofline.frac_excluding_tail) is computed from."""
    from pympc_amd import fixtures
    from test_gpu_parity import _stacked_batch
    B, K_STEPS = 40, 8
    K = _stacked_batch([fixtures.random_lti(2000 + i) for i in range(B)]); K.setup()
    bp = K.prob
    bp.profile(enable=True, reset=True)
    tr = K.run(K_STEPS)
    ms, launches = bp.profile(enable=False)
    t = bp.launch_times(K_STEPS).astype(np.int64)
    assert t.shape == (B, 2 + K_STEPS) and launches == 1
    seq = np.concatenate([t[:, :1], t[:, 2:], t[:, 1:2]], axis=1)          # entry, step ends, exit
    assert (np.diff(seq, axis=1) >= 0).all() and (t[:, 0] > 0).all()
    span_ms = (t[:, 1].max() - t[:, 0].min()) * 1e-5                       # 10 ns ticks
    assert 0.0 < span_ms <= ms * 1.05 + 0.05, (span_ms, ms)
    assert (tr['status'] == 1).all()
    with pytest.raises(RuntimeError):
        bp.launch_times(65)                                                 # at most 64 step stamps are kept
