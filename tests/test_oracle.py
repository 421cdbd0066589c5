"""CPU oracle (oracle/osqp_ref.c) pinned against the golden vectors:
  * it solves the reference-built QPs (tests/golden/qp_*.npz, captured from the imported reference) to the
    certified optimum stored in tests/golden/opt_*.npz,
  * the optimum satisfies the KKT conditions on the reference-built matrices (solver-independent certificate),
  * at the reference's default tolerance it reports OSQP's status strings and check-interval granularity.
"""
import warnings

import numpy as np
import pytest

from util import golden_names, load_golden, golden_csc, golden_kwargs, kkt_certificate, update_steps, apply_attrs
from oracle.osqp_oracle import OSQP


@pytest.mark.parametrize('name', golden_names())
def test_oracle_reaches_certified_optimum(name):
    g, opt = load_golden(name), load_golden(name, prefix='opt_')
    P, A = golden_csc(g, 'P'), golden_csc(g, 'A')
    prob = OSQP()
    prob.setup(P, g['q'], A, g['l'], g['u'], eps_abs=1e-11, eps_rel=1e-11, max_iter=400000)
    r = prob.solve()
    assert r.info.status == 'solved'
    stat, pv, comp = kkt_certificate(P, g['q'], A, g['l'], g['u'], r.x, r.y)
    assert stat < 1e-8 and pv < 1e-8 and comp < 1e-8
    assert np.abs(r.x - opt['x']).max() <= 1e-7 * max(1.0, np.abs(opt['x']).max())
    if 'highs_rel_diff_u0' in opt.files:            # independent QP solver agreed when the golden was made
        assert float(opt['highs_rel_diff_u0']) < 1e-5


@pytest.mark.parametrize('name', ['point_mass', 'cart_pole', 'small_mimo', 'point_mass_nc'])
def test_oracle_default_tolerance_and_warm_start(name):
    from pympc_amd import MPCController
    g = load_golden(name)
    K = apply_attrs(MPCController(**golden_kwargs(g)), golden_kwargs(g))
    K.prob = OSQP()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup()
    assert K.res.info.status == 'solved'
    assert K.res.info.iter % 25 == 0 and K.res.info.iter > 0        # check_termination = 25
    cold = K.res.info.iter
    st = update_steps(g)[0]
    K.update(K.x0 + 1e-3, u=K.output())
    assert K.res.info.status == 'solved' and K.res.info.iter <= cold  # warm start from the previous iterate


def test_oracle_update_rejects_crossed_bounds():
    g = load_golden('point_mass')
    prob = OSQP()
    prob.setup(golden_csc(g, 'P'), g['q'], golden_csc(g, 'A'), g['l'], g['u'])
    with pytest.raises(ValueError):
        prob.update(l=g['u'] + 1.0, u=g['u'])


def test_alongside_worker_flags_what_differs_and_nothing_else():
    """oracle/cpu_bench.alongside_pool (what tests/test_gpu_gaps.py checks the headline batch's iteration counts with, on every instance) on a
    trajectory the oracle itself produced: no difference; with one count and one status tampered with: exactly those."""
    from pympc_amd import MPCController, fixtures
    from oracle import cpu_bench
    nx, nu, Np, xbox, B, K = 5, 2, 8, 10.0, 3, 4
    tr = dict(x=np.zeros((K + 1, B, nx)), u=np.zeros((K, B, nu)), iter=np.zeros((K, B), dtype=np.int32), status=np.zeros((K, B), dtype=np.int32))
    for i in range(B):
        kw = fixtures.random_lti(i, nx=nx, nu=nu, Np=Np, xbox=xbox); kw.update(eps_abs=1e-3, eps_rel=1e-3)
        Ko = MPCController(**kw); Ko.prob = OSQP()
        rng = fixtures.random_lti_noise_rng(i)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            Ko.setup()
            x = np.array(kw['x0'], dtype=float)
            tr['x'][0, i] = x
            for k in range(K):
                u = Ko.output()
                x = kw['Ad'] @ x + kw['Bd'] @ u + 0.01 * rng.standard_normal(nx)
                Ko.update(x, u)
                tr['u'][k, i], tr['x'][k + 1, i], tr['iter'][k, i], tr['status'][k, i] = u, x, Ko.res.info.iter, Ko.res.info.status_val
    res = cpu_bench.alongside_pool(np.arange(B), tr, 1e-3, nx, nu, Np, xbox, workers=2)
    assert [r[0] for r in res] == [0, 1, 2]
    assert all(not r[2] and not r[3] and r[4] <= 1e-12 for r in res), res
    assert sum(r[5] for r in res) == int(tr['iter'].sum())
    tr['iter'][2, 1] += 25; tr['status'][3, 2] = 2
    res = cpu_bench.alongside_pool(np.arange(B), tr, 1e-3, nx, nu, Np, xbox, workers=2)
    assert res[0][2] == [] and res[0][3] == []
    assert [k for k, _, _ in res[1][2]] == [2] and res[1][3] == []
    assert res[2][2] == [] and [k for k, _, _ in res[2][3]] == [3]
