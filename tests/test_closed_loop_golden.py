"""Closed-loop golden trajectories produced by the REFERENCE's controller class (tests/golden/make_traj.py): our host
class, driven through the same caller loop with the CPU oracle as its solver, must visit the same states and apply
the same inputs -- QP refresh, warm starts, u_{-1} bookkeeping and output() over whole runs (mpc.py:271-364,386-454)."""
import warnings

import numpy as np
import pytest

from util import traj_names, load_traj, cart_pole_plant


def _closed_loop(K, kw, g, plant, feed_golden_states, tol):
    pattern = str(g['pattern'])
    xs, us = g['x'], g['u']
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        K.setup()
        x, u = np.array(kw['x0'], dtype=float), np.array(kw['uminus1'], dtype=float)
        for k in range(len(us)):
            if pattern == 'update_output':
                K.update(x, u)
                u = K.output()
            else:
                u = K.output()
            scale = max(1e-3, np.abs(us).max())
            assert np.abs(u - us[k]).max() <= tol * scale, (k, u, us[k])
            x = xs[k + 1] if feed_golden_states else plant(x, u)
            assert np.abs(x - xs[k + 1]).max() <= tol * max(1e-3, np.abs(xs).max()), k
            if pattern != 'update_output':
                K.update(x)


@pytest.mark.parametrize('name', traj_names())
def test_host_class_with_oracle_reproduces_reference_closed_loop(name):
    from pympc_amd import MPCController, fixtures
    from oracle.osqp_oracle import OSQP
    g = load_traj(name)
    kw = dict(fixtures.NAMED[str(g['fixture'])]())
    kw.update(eps_abs=float(g['eps']), eps_rel=float(g['eps']))
    K = MPCController(**kw)
    K.prob = OSQP()
    K.solver_settings = dict(max_iter=400000)
    Ad, Bd = kw['Ad'], kw['Bd']
    plant = cart_pole_plant if name == 'cart_pole' else (lambda x, u: Ad @ x + Bd @ u)
    _closed_loop(K, kw, g, plant, feed_golden_states=False, tol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize('name', traj_names())
def test_gpu_controller_follows_reference_closed_loop(name):
    """The drop-in class on the GPU (eps 1e-9) along the reference's trajectory: every applied input within 1e-6 (relative
    to the input range) of the reference controller's -- the north-star criterion, step after step."""
    from pympc_amd import MPCController, fixtures
    g = load_traj(name)
    kw = dict(fixtures.NAMED[str(g['fixture'])]())
    kw.update(eps_abs=1e-9, eps_rel=1e-9)
    K = MPCController(**kw)
    K.solver_settings = dict(max_iter=400000)
    _closed_loop(K, kw, g, None, feed_golden_states=True, tol=1e-6)


@pytest.mark.parametrize('name', traj_names(output_feedback=True))
def test_host_estimator_and_class_reproduce_reference_output_feedback_loop(name):
    """pympc_amd.kalman.LinearStateEstimator next to our MPCController (oracle as solver) against the loop the REFERENCE's
    LinearStateEstimator + MPCController ran (make_traj.py: examples/example_inverted_pendulum_kalman.py:135-174 with a
    linear plant and recorded noise): estimates, inputs and states to 1e-8 -- pins pyMPC/kalman.py:109-134's
    predict/update and what update(KF.x, u) does with the estimate."""
    from pympc_amd import MPCController, fixtures
    from pympc_amd.kalman import LinearStateEstimator
    from oracle.osqp_oracle import OSQP
    g = load_traj(name)
    kw = dict(fixtures.NAMED[str(g['fixture'])]())
    kw.update(eps_abs=float(g['eps']), eps_rel=float(g['eps']))
    Ad, Bd, Cd, L = kw['Ad'], kw['Bd'], g['C'], g['L']
    K = MPCController(**kw); K.prob = OSQP(); K.solver_settings = dict(max_iter=400000)
    KF = LinearStateEstimator(np.array(kw['x0'], dtype=float), Ad, Bd, Cd, np.zeros((Cd.shape[0], Bd.shape[1])), L)
    x = np.array(g['x_true0'], dtype=float)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        K.setup()
        for k in range(len(g['u'])):
            y = Cd @ x + g['v'][k]
            u = K.output()
            x = Ad @ x + Bd @ u + g['w'][k]
            KF.update(y); KF.predict(u)
            K.update(KF.x, u)
            assert np.abs(u - g['u'][k]).max() <= 1e-8 * max(1e-3, np.abs(g['u']).max()), k
            assert np.abs(x - g['x'][k + 1]).max() <= 1e-8 * np.abs(g['x']).max(), k
            assert np.abs(KF.x - g['xhat'][k + 1]).max() <= 1e-8 * np.abs(g['xhat']).max(), k


def _no_slack_loop(K, g, tol):
    """u = K.step(); x+ = Ad x + Bd u; K.update(x)  -- the loop of mpc_no_slack.py:362-371."""
    xs, us = g['x'], g['u']
    K.setup()
    x = xs[0].copy()
    for k in range(len(us)):
        u = K.step()
        assert np.abs(u - us[k]).max() <= tol * max(1e-3, np.abs(us).max()), (k, u, us[k])
        x = K.Ad @ x + K.Bd @ u
        assert np.abs(x - xs[k + 1]).max() <= tol * np.abs(xs).max(), k
        K.update(x)


def _no_slack_controller(g):
    from pympc_amd import fixtures
    from pympc_amd.mpc_no_slack import MPCController
    kw = {k: v for k, v in fixtures.point_mass().items() if k != 'eps_feas'}
    kw.update(xmin=g['xmin'], xmax=g['xmax'])
    K = MPCController(**kw)
    K.eps_abs = K.eps_rel = float(g['eps'])          # (the golden run was made at tight tolerance; the class's own is 1e-4)
    K.solver_settings = dict(max_iter=400000)
    return K


def test_no_slack_shim_with_oracle_reproduces_reference_class():
    """pympc_amd.mpc_no_slack.MPCController (oracle as its solver) against the trajectory the REFERENCE's
    pyMPC/mpc_no_slack.py class produced in the loop of its own __main__ (make_traj.py): hard state box, step()/update()."""
    from oracle.osqp_oracle import OSQP
    g = load_traj('no_slack_point_mass')
    K = _no_slack_controller(g)
    K.prob = OSQP()
    _no_slack_loop(K, g, 1e-8)
    assert K.P.shape[0] == (K.Np + 1) * K.nx + K.Np * K.nu           # no slack columns
    K2 = _no_slack_controller(g); K2.prob = OSQP(); K2.setup()
    K2.update(np.array([0.1, 0.2]), np.array([9.0]))                  # u_{-1} far outside the input box: infeasible -> step() raises
    with pytest.raises(ValueError, match='OSQP did not solve the problem!'):
        K2.step()


def test_no_slack_shim_rejects_more_than_one_input():
    from pympc_amd import fixtures
    from pympc_amd.mpc_no_slack import MPCController
    kw = {k: v for k, v in fixtures.accel_brake().items() if k != 'eps_feas'}
    with pytest.raises(ValueError, match='single input'):
        MPCController(**kw).setup()


@pytest.mark.gpu
def test_no_slack_shim_on_gpu_follows_reference_class():
    g = load_traj('no_slack_point_mass')
    _no_slack_loop(_no_slack_controller(g), g, 1e-6)
