"""mpcqp_share_factor / the map every setup builds: one model, many states (test_scripts/example_mpc_function.py:105-111; SURVEY 8(e): broadcast the model, scatter only x0).  Instances
whose factorization inputs equal instance 0's solve with ONE shared copy of its factor.  Sharing must not change a single bit of any result -- not when an
instance adapts rho in the middle of a launch and leaves the shared slot, not in the closed loop -- and it must not share what is not identical."""
import warnings

import numpy as np
import pytest

from pympc_amd import _lib

pytestmark = pytest.mark.gpu


def _setup(B, kw, backend, x0=None, **settings):
    from pympc_amd.solver import BatchProblem
    nx, nu = kw['Ad'].shape[0], kw['Bd'].shape[1]
    prob = BatchProblem(B, nx, nu, kw['Np'], backend=backend, warm_start=1, **settings)
    bc = lambda v: np.broadcast_to(np.asarray(v, dtype=float), (B,) + np.shape(v))
    x0 = bc(kw['x0']) if x0 is None else x0
    prob.setup(bc(kw['Ad']), bc(kw['Bd']), bc(kw['Qx']), bc(kw['QxN']), bc(kw['Qu']), bc(kw['QDu']), bc(kw['xmin']), bc(kw['xmax']), bc(kw['umin']),
               bc(kw['umax']), bc(kw['Dumin']), bc(kw['Dumax']), bc(kw['uref']), np.full((B, 1), kw['eps_feas']), x0, bc(kw['uminus1']), bc(kw['xref']))
    return prob


def _walk(prob, X0, W, steps):
    """cold solve at the common state, scattered states, a warm solve, then a closed loop: everything observable"""
    B = X0.shape[0]
    prob.solve_async(); prob.synchronize()
    out = [prob.solution()[0].copy()]
    prob.update(X0, np.zeros((B, prob.nu)))
    prob.solve_async(); prob.synchronize()
    x, y, info = prob.solution()
    out += [x.copy(), y.copy(), np.array([(i.status, i.iter, i.rho_updates) for i in info]), np.array([i.rho for i in info])]
    xt, ut, st, it = prob.mpc_run(steps, w=W)
    out += [xt, ut, st, it]
    return out, prob.stats()


# (the last: more instances than resident workgroup slots -- the closed loop runs persistently, (instance, step range) items off a queue, and an instance
#  that left the shared slot in one item is picked up by another workgroup for the next)
@pytest.mark.parametrize('shape', [(12, 4, 30, 10.0, 96), (20, 8, 40, 1.0, 24), (4, 2, 60, 10.0, 40), (12, 4, 30, 1.0, 1300)])
def test_shared_factor_changes_no_bit(shape):
    from pympc_amd import fixtures
    nx, nu, Np, xbox, B = shape
    kw = fixtures.random_lti(3, nx=nx, nu=nu, Np=Np, xbox=xbox)
    rng = np.random.default_rng(11)
    X0 = kw['x0'][None, :] * rng.uniform(0.3, 2.5, size=(B, 1)) * rng.choice([-1.0, 1.0], size=(B, nx))      # some states far out: those instances adapt rho
    W = 0.01 * rng.standard_normal((12, B, nx))
    res = []
    for share in (False, True):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            # (tight: solves long enough for OSQP's rho adaptation to act; setup shares by itself unless told not to)
            prob = _setup(B, kw, 'sweeps', eps_abs=1e-8, eps_rel=1e-8, max_iter=20000, tuning=0 if share else _lib.TUNE_NO_SHARE)
            prob.solve_async(); prob.synchronize()
            if share:
                assert prob.share_factor() == B                  # (the map again, against instance 0's factor after the cold solve: identical instances adapted alike)
            res.append(_walk(prob, X0, W, 12))
            prob.close()
    (a, sa), (b, sb) = res
    assert sa == sb                                              # iterations, checks, refactorizations, solves
    print(shape, 'stats', sa)
    assert sa[2] > 0                                             # the walk does contain rho updates: instances LEFT the shared slot and carried on
    for u, v in zip(a, b):
        assert np.array_equal(u, v)


def test_only_identical_instances_share_and_setup_ends_it():
    from pympc_amd import fixtures
    from pympc_amd.solver import BatchProblem
    B = 16
    kw = fixtures.random_lti(5)
    bc = lambda v: np.broadcast_to(np.asarray(v, dtype=float), (B,) + np.shape(v)).copy()
    Bd, x0 = bc(kw['Bd']), bc(kw['x0'])
    Bd[5, 0, 0] = np.nextafter(Bd[5, 0, 0], 10.0)                # one ulp in the model: not the same factorization any more
    x0[9] *= 0.5                                                 # the state at setup enters the bounds only, not the factor: instance 9 shares

    def make(tuning=0):
        prob = BatchProblem(B, 12, 4, kw['Np'], backend='sweeps', warm_start=1, tuning=tuning)
        prob.setup(bc(kw['Ad']), Bd, bc(kw['Qx']), bc(kw['QxN']), bc(kw['Qu']), bc(kw['QDu']), bc(kw['xmin']), bc(kw['xmax']), bc(kw['umin']),
                   bc(kw['umax']), bc(kw['Dumin']), bc(kw['Dumax']), bc(kw['uref']), np.full((B, 1), kw['eps_feas']), x0, bc(kw['uminus1']), bc(kw['xref']))
        return prob

    prob, ref = make(), make(_lib.TUNE_NO_SHARE)
    assert prob.share_factor() == B - 1
    for p in (prob, ref):
        p.solve_async(); p.synchronize()
    assert np.array_equal(prob.solution()[0], ref.solution()[0])
    # every setup call factors every instance into its own slot and builds the map anew
    prob.setup(bc(kw['Ad']), bc(kw['Bd']), bc(kw['Qx']), bc(kw['QxN']), bc(kw['Qu']), bc(kw['QDu']), bc(kw['xmin']), bc(kw['xmax']), bc(kw['umin']),
               bc(kw['umax']), bc(kw['Dumin']), bc(kw['Dumax']), bc(kw['uref']), np.full((B, 1), kw['eps_feas']), x0, bc(kw['uminus1']), bc(kw['xref']))
    assert prob.share_factor() == B
    ref.close(); prob.close()


def test_shared_batch_against_the_oracle():
    """The one-model-many-states batch on ONE shared factor against the CPU oracle (oracle/osqp_ref.c), state by state at the north-star tolerance."""
    from pympc_amd import fixtures, MPCController
    from oracle.osqp_oracle import OSQP
    B = 64
    kw = fixtures.random_lti(7)
    rng = np.random.default_rng(3)
    X0 = rng.standard_normal((B, 12)) * rng.uniform(0.2, 3.0, size=(B, 1))
    Um1 = rng.uniform(-0.5, 0.5, size=(B, 4))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prob = _setup(B, kw, 'sweeps', eps_abs=1e-9, eps_rel=1e-9, max_iter=200000)
        assert prob.share_factor() == B
        prob.update(X0, Um1)
        prob.solve_async(); prob.synchronize()
        U = prob.u0()
        assert all(i.status == 1 for i in prob.infos())
        for i in range(0, B, 7):
            Ko = MPCController(**dict(kw, x0=X0[i], uminus1=Um1[i], eps_abs=1e-9, eps_rel=1e-9)); Ko.prob = OSQP(); Ko.solver_settings = dict(max_iter=200000); Ko.setup()
            uo = Ko.output()
            assert np.abs(U[i] - uo).max() <= 1e-6 * max(1e-3, np.abs(uo).max()), (i, U[i], uo)
        prob.close()


def test_register_resident_backends_have_nothing_to_share():
    from pympc_amd import fixtures
    kw = fixtures.random_lti(5)
    prob = _setup(8, kw, 'bcr8')
    assert prob.share_factor() == 0
    prob.solve_async(); prob.synchronize()
    assert all(i.status == 1 for i in prob.infos())
    prob.close()
