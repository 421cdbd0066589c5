"""The solver seam exactly as pyMPC uses it (mpc.py:241,266,454,369): ``prob.setup(P, q, A, l, u, **settings)``,
``prob.update(l=, u=, q=)``, ``prob.solve()`` with the CALLER's matrices and vectors -- here the ones the imported
reference class built (tests/golden/qp_*.npz, including its three scripted update() calls) -- against the oracle given the
very same objects.  This is what a maintainer gets by replacing ``osqp.OSQP()`` with ``pympc_amd.solver.DeviceProblem()``
in pyMPC/mpc.py and nothing else."""
import warnings

import numpy as np
import pytest

from util import golden_names, load_golden, golden_csc, update_steps

pytestmark = pytest.mark.gpu


def _both(g, **settings):
    from pympc_amd.solver import DeviceProblem
    from oracle.osqp_oracle import OSQP
    P, A = golden_csc(g, 'P'), golden_csc(g, 'A')
    pd, po = DeviceProblem(), OSQP()
    pd.setup(P, g['q'], A, g['l'], g['u'], warm_start=True, verbose=False, **settings)
    po.setup(P, g['q'], A, g['l'], g['u'], warm_start=True, verbose=False, **settings)
    return pd, po


@pytest.mark.parametrize('name', golden_names())          # (the *_hard fixtures: SOFT_ON = False, recognised by the library from the pattern)
def test_reference_shaped_calls_at_default_tolerance(name):
    """eps 1e-3 (mpc.py:80): same status, iteration count and iterate after setup and after every update(l, u, q)."""
    g = load_golden(name)
    pd, po = _both(g, eps_abs=1e-3, eps_rel=1e-3)
    steps = [None] + update_steps(g)
    for st in steps:
        if st is not None:
            pd.update(l=st['l'], u=st['u_bound'], q=st['q']); po.update(l=st['l'], u=st['u_bound'], q=st['q'])
        rd, ro = pd.solve(), po.solve()
        assert rd.info.status == ro.info.status
        # same number of iterations -- or, when a residual sits on the tolerance at a termination check (both solvers
        # evaluate it to ~1e-12 of each other and may land on different sides), one check interval apart.  The two
        # chains of warm starts have then parted: the remaining steps of the fixture are not comparable iterate by iterate.
        if rd.info.iter != ro.info.iter:
            assert abs(rd.info.iter - ro.info.iter) == 25 and st is not None      # (never on the cold solve)
            break
        # The iterate itself: not the optimum but a point within eps = 1e-3 of it, reached through a chain of warm-started
        # solves by two different factorizations (block LDL' with explicit S^-1 here, sparse LDL' of the quasi-definite KKT
        # there).  Measured: 1e-12 (12,4,30) ... 1.3e-5 (20,8,12 with active slack rows, after three warm solves; 4e-7 with
        # the two-slot factor format) -- compared at a tenth of the solver tolerance.
        assert np.abs(rd.x - ro.x).max() <= 1e-4 * max(1.0, np.abs(ro.x).max())
        # (the objective of such a point is only determined to O(eps |obj|): the weights amplify the iterate's spread -- 1.1e-4
        #  relative measured on (5,3,8) -- so it is compared at the solver tolerance itself)
        assert abs(rd.info.obj_val - ro.info.obj_val) <= 1e-3 * max(1.0, abs(ro.info.obj_val))


@pytest.mark.parametrize('name', golden_names())
def test_reference_shaped_calls_reach_the_certified_optimum(name):
    g, opt = load_golden(name), load_golden(name, prefix='opt_')
    pd, _ = _both(g, eps_abs=1e-11, eps_rel=1e-11, max_iter=400000)
    r = pd.solve()
    assert r.info.status == 'solved'
    assert np.abs(r.x - opt['x']).max() <= 1e-6 * np.abs(opt['x']).max()
    nx, nu, Np = int(g['in_Ad'].shape[0]), int(g['in_Bd'].shape[1]), int(g['in_Np'])
    u0 = r.x[(Np + 1) * nx:(Np + 1) * nx + nu]
    assert np.abs(u0 - opt['u0']).max() <= 1e-6 * max(1e-3, np.abs(opt['u0']).max())


def test_partial_updates_and_return_to_device_built_vectors():
    """update(q=...) alone / update(l=, u=) alone (osqp allows either); and a controller that set the problem up from its
    data (mpc=) may still be handed raw vectors later, then go back to device-built ones."""
    from pympc_amd import MPCController, fixtures
    from oracle.osqp_oracle import OSQP
    g = load_golden('random_12_4_30')
    pd, po = _both(g, eps_abs=1e-9, eps_rel=1e-9, max_iter=100000)
    st = update_steps(g)
    pd.update(q=st[2]['q']); po.update(q=st[2]['q'])
    assert np.abs(pd.solve().x - po.solve().x).max() <= 1e-6
    pd.update(l=st[0]['l'], u=st[0]['u_bound']); po.update(l=st[0]['l'], u=st[0]['u_bound'])
    assert np.abs(pd.solve().x - po.solve().x).max() <= 1e-6
    kw = dict(fixtures.random_lti(5)); kw.update(eps_abs=1e-9, eps_rel=1e-9)
    K = MPCController(**kw); K.solver_settings = dict(max_iter=100000)
    Ko = MPCController(**kw); Ko.prob = OSQP(); Ko.solver_settings = dict(max_iter=100000)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        K.setup(); Ko.setup()
        x1 = kw['Ad'] @ kw['x0'] + kw['Bd'] @ Ko.output()
        K.output()
        Ko.update(x1)                                    # reference-style refresh on the host ...
        K.prob._bp.update_vectors(Ko.q[None], np.clip(Ko.l, -1e30, 1e30)[None], np.clip(Ko.u, -1e30, 1e30)[None])   # ... handed over raw
        K.res = K.prob.solve()
        assert np.abs(K.res.x - Ko.res.x).max() <= 1e-6 * np.abs(Ko.res.x).max()
        x2 = kw['Ad'] @ x1 + kw['Bd'] @ Ko.output()
        K.uminus1_rh = Ko.uminus1_rh
        K.update(x2); Ko.update(x2)                      # back to vectors built on the device
        assert np.abs(K.output() - Ko.output()).max() <= 1e-6
