"""The reference's Monte-Carlo loop over ONE controller (test_scripts/example_mpc_function.py:105-111: 10 000 random (x, u_{-1}) through
K.__controller_function__) as ONE batched evaluation (examples/controller_map_monte_carlo.py): every state an instance of a BatchMPCController that
broadcasts the model.  At a tight tolerance the batched map, the single controller called state by state and the CPU oracle agree."""
import os
import sys
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'examples'))


def test_batched_controller_map_matches_the_sequential_calls_and_the_oracle():
    import controller_map_monte_carlo as ex
    from pympc_amd import MPCController
    from oracle.osqp_oracle import OSQP
    rng = np.random.default_rng(5)
    n = 1500
    X, Um1 = rng.random((n, 2)), rng.random((n, 1))
    kw = ex.point_mass(1e-9)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        U, Kb = ex.controller_map(kw, X, Um1, max_iter=200000)
        assert all(inf.status == 1 for inf in Kb.prob.infos())                                  # every instance solved
        K = MPCController(x0=X[0], uminus1=Um1[0], **kw); K.solver_settings = dict(max_iter=200000); K.setup()
        Ko = MPCController(x0=X[0], uminus1=Um1[0], **kw); Ko.prob = OSQP(); Ko.solver_settings = dict(max_iter=200000); Ko.setup()
        for i in range(0, n, 25):
            u1 = K.__controller_function__(X[i], Um1[i])            # warm-started from the previous, unrelated state: as the reference loop
            uo = Ko.__controller_function__(X[i], Um1[i])
            assert np.abs(U[i] - uo).max() <= 1e-6 * max(1e-3, np.abs(uo).max()), (i, U[i], uo)
            assert np.abs(u1 - uo).max() <= 1e-6 * max(1e-3, np.abs(uo).max()), (i, u1, uo)


def test_one_controller_many_states_on_a_shared_factor():
    """The reference loop's own order -- ONE controller set up, then only the state moves -- as a batch of copies of that controller on one shared KKT
    factor (forced streaming backend: the library's choice for a few hundred of these tiny problems is the register-resident inverse, which has nothing to share)."""
    import controller_map_monte_carlo as ex
    rng = np.random.default_rng(6)
    n = 700
    X, Um1 = rng.random((n, 2)), rng.random((n, 1))
    kw = ex.point_mass(1e-9)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        U, _ = ex.controller_map(kw, X, Um1, max_iter=200000)
        U1, K1, sharing = ex.controller_map_one_controller(kw, X, Um1, X[0], Um1[0], max_iter=200000, backend='sweeps')
    assert sharing == n
    assert all(inf.status == 1 for inf in K1.prob.infos())
    assert np.abs(U1 - U).max() <= 1e-6 * max(1e-3, np.abs(U).max())
