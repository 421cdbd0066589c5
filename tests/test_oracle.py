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
