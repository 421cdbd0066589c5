#!/usr/bin/env python3
"""Closed-loop golden trajectories: the REFERENCE's own controller class -- imported from
/root/reference, its `import osqp` satisfied by a stub module that hands it the CPU oracle at tight tolerance -- is
driven through the caller loops of the reference's example scripts, and the visited states, applied inputs and solver
statuses are stored.  What is pinned is everything the reference class does around the solver over a whole run:
QP construction, the per-step q/l/u refresh (mpc.py:386-454), warm starts, u_{-1} bookkeeping and output()
(mpc.py:271-336).  The optimum of each step's QP is unique, so (x_k, u_k) do not depend on the solver used.

Loops (inputs are the constants of pympc_amd/fixtures.py, cited there):
  point_mass, accel_brake, quadcopter : K.update(x, u); u = K.output(); x+ = Ad x + Bd u
        (examples/example_point_mass.py:88-101 with the exact discrete step of mpc.py:690 instead of the ODE integrator)
  cart_pole                           : same calls, nonlinear plant + forward Euler of examples/example_inverted_pendulum.py:83-103
  point_mass_nc                       : u = K.output(); x+ = Ad x + Bd u; K.update(x)   (mpc.py:688-692, 2-D xref, Nc < Np)
  no_slack_point_mass                 : the REFERENCE's older class pyMPC/mpc_no_slack.py (hard state box) through the loop of its own
        __main__ (mpc_no_slack.py:362-371): u = K.step(); x+ = Ad x + Bd u; K.update(x), constants of mpc_no_slack.py:296-349
        (= the point-mass fixture with the state box [-10,-10]..[7,10]); tolerance as hard-coded there (1e-4) AND tight.
  kalman_cart_pole                    : output feedback, the loop of examples/example_inverted_pendulum_kalman.py:135-174 with
        the REFERENCE's LinearStateEstimator (pyMPC/kalman.py:109-134) next to its MPCController, linear plant and recorded
        noise:  y = C x + v; u = K.output(); x+ = Ad x + Bd u + w; KF.update(y); KF.predict(u); K.update(KF.x, u).
        pyMPC/kalman.py imports `control` at module top (absent here; only its design functions use it) and calls the
        long-removed alias `scipy.size` in LinearStateEstimator.__init__: the script satisfies the import with an empty
        module and restores the alias (= numpy.size, what it was) for the duration of the run.  The filter gain L is
        INPUT data (designed by pympc_amd.kalman.kalman_design_simple) and is stored with the trajectory.

  kalman_cart_pole_np200              : the same loop with the example's own controller and estimator settings (Ts = 5 ms, Np = Nc = 200,
        eps_feas = 1e3, Q_kal = 10 I, R_kal = I: example_inverted_pendulum_kalman.py:19,100-110), fixture cart_pole_kalman.

    python tests/golden/make_traj.py [name ...]   # needs /root/reference; writes tests/golden/traj_<name>.npz
"""
import os
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

STUB = '''
import sys
sys.path.insert(0, %r)
from oracle.osqp_oracle import OSQP as _Oracle
TIGHT = None
class OSQP(_Oracle):
    def setup(self, *a, **kw):
        kw.setdefault('max_iter', 400000)
        if TIGHT is not None:
            kw['eps_abs'] = kw['eps_rel'] = TIGHT
        return super().setup(*a, **kw)
''' % REPO

EPS = 1e-10


def cart_pole_plant(x, u, Ts=50e-3):
    """examples/example_inverted_pendulum.py:10-17,92-103 (constants and forward-Euler step of the nonlinear model)."""
    M, m, b, ftheta, l, g = 0.5, 0.2, 0.1, 0.1, 0.3, 9.81
    F, v, theta, omega = float(u[0]), x[1], x[2], x[3]
    der = np.zeros(4)
    der[0] = v
    der[1] = (m * l * np.sin(theta) * omega ** 2 - m * g * np.sin(theta) * np.cos(theta) + m * ftheta * np.cos(theta) * omega + F - b * v) / (M + m * (1 - np.cos(theta) ** 2))
    der[2] = omega
    der[3] = ((M + m) * (g * np.sin(theta) - ftheta * omega) - m * l * omega ** 2 * np.sin(theta) * np.cos(theta) - (F - b * v) * np.cos(theta)) / (l * (M + m * (1 - np.cos(theta) ** 2)))
    return x + der * Ts


CASES = {   # name -> (fixture, steps, loop pattern)
    'point_mass': ('point_mass', 75, 'update_output'),
    'cart_pole': ('cart_pole', 120, 'update_output'),
    'accel_brake': ('accel_brake', 60, 'update_output'),
    'quadcopter': ('quadcopter', 40, 'update_output'),
    'point_mass_nc': ('point_mass_nc', 40, 'output_update'),
}


def run(Ctrl, kw, steps, pattern, plant):
    K = Ctrl(**kw)
    with warnings.catch_warnings():
        warnings.simplefilter('error')          # a non-'solved' step would fall back to u_failure: never in these runs
        K.setup()
        x = np.array(kw['x0'], dtype=float)
        u = np.array(kw['uminus1'], dtype=float)
        xs, us = [x.copy()], []
        for _ in range(steps):
            if pattern == 'update_output':
                K.update(x, u)
                u = K.output()
                x = plant(x, u)
            else:
                u = K.output()
                x = plant(x, u)
                K.update(x)
            us.append(np.array(u, dtype=float)); xs.append(np.array(x, dtype=float))
    return np.array(xs), np.array(us)


def run_kalman(Ctrl, Estimator, kw, steps, Q_kal=None, R_kal=None, vstd=0.01, wstd=0.001):
    """Output-feedback closed loop with the reference's own estimator class; returns everything the replay needs."""
    from pympc_amd.kalman import kalman_design_simple
    Ad, Bd = kw['Ad'], kw['Bd']
    nx, nu = Bd.shape
    Cd = np.array([[1.0, 0, 0, 0], [0, 0, 1.0, 0]])            # position and angle are measured (example :62-66)
    Dd = np.zeros((2, nu))
    Q_kal = np.diag([0.1, 10, 0.1, 10]) if Q_kal is None else Q_kal
    R_kal = 0.01 * np.eye(2) if R_kal is None else R_kal
    L, _, _ = kalman_design_simple(Ad, Bd, Cd, Dd, Q_kal, R_kal, type='filter')
    rng = np.random.default_rng(77)
    v = vstd * rng.standard_normal((steps, 2))
    w = wstd * rng.standard_normal((steps, nx))
    x = np.array(kw['x0'], dtype=float) * 1.05                 # the controller starts from a wrong estimate
    x_true0 = x.copy()
    KF = Estimator(np.array(kw['x0'], dtype=float), Ad, Bd, Cd, Dd, L)
    K = Ctrl(**kw)
    xs, xh, us, ys = [x.copy()], [np.array(KF.x, dtype=float)], [], []
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        K.setup()
        for k in range(steps):
            y = Cd @ x + v[k]
            u = K.output()
            x = Ad @ x + Bd @ u + w[k]
            KF.update(y)
            KF.predict(u)
            K.update(KF.x, u)
            xs.append(x.copy()); xh.append(np.array(KF.x, dtype=float)); us.append(np.array(u, dtype=float)); ys.append(y)
    return dict(x=np.array(xs), xhat=np.array(xh), u=np.array(us), y=np.array(ys), v=v, w=w, C=Cd, L=L, x_true0=x_true0)


def main():
    from pympc_amd import fixtures
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, 'osqp'))
        with open(os.path.join(tmp, 'osqp', '__init__.py'), 'w') as f:
            f.write(STUB)
        os.makedirs(os.path.join(tmp, 'control'))
        with open(os.path.join(tmp, 'control', '__init__.py'), 'w') as f:
            f.write('# placeholder: pyMPC/kalman.py imports `control` at module top; the estimator class never calls it\n')
        sys.path.insert(0, tmp)
        sys.path.insert(0, '/root/reference')
        from pyMPC.mpc import MPCController as RefController
        if not sys.argv[1:] or 'no_slack_point_mass' in sys.argv[1:]:
            import pyMPC.mpc_no_slack as ref_ns
            kw = {k: v for k, v in fixtures.point_mass().items() if k != 'eps_feas'}
            kw.update(xmin=np.array([-10.0, -10.0]), xmax=np.array([7.0, 10.0]))
            K = ref_ns.MPCController(**kw)
            # mpc_no_slack.py:119 hard-codes eps 1e-4 in its setup call; the stub solver is told to use the tight tolerance instead
            # (the optimum is what is stored; the shim's own 1e-4 behaviour is compared with the oracle at 1e-4 in the tests)
            ref_ns.osqp.TIGHT = EPS
            K.setup()
            x = np.array(kw['x0'], dtype=float); xs, us = [x.copy()], []
            for _ in range(40):                        # (stops short of the position bound: ON a hard bound x0 = 7 + 1e-10 makes the
                                                       #  next QP infeasible at this tolerance and the reference raises -- why mpc.py has slack)
                u = K.step(); x = kw['Ad'] @ x + kw['Bd'] @ u; K.update(x)
                xs.append(x.copy()); us.append(np.array(u, dtype=float))
            ref_ns.osqp.TIGHT = None
            np.savez_compressed(os.path.join(HERE, 'traj_no_slack_point_mass.npz'), x=np.array(xs), u=np.array(us), pattern='step_update', fixture='point_mass', eps=EPS,
                                xmin=kw['xmin'], xmax=kw['xmax'])
            print('%-14s %3d steps  |x|max %.3f  |u|max %.3f  u[0] %s' % ('no_slack_point_mass', 40, np.abs(xs).max(), np.abs(us).max(), us[0]))
        if not sys.argv[1:] or 'kalman_cart_pole' in sys.argv[1:]:
            import scipy
            if not hasattr(scipy, 'size'):
                scipy.size = np.size                            # the alias kalman.py:9-20 was written against
            from pyMPC.kalman import LinearStateEstimator as RefEstimator
            kw = dict(fixtures.NAMED['cart_pole']())
            kw.update(eps_abs=EPS, eps_rel=EPS)
            out = run_kalman(RefController, RefEstimator, kw, 40)
            np.savez_compressed(os.path.join(HERE, 'traj_kalman_cart_pole.npz'), pattern='output_feedback', fixture='cart_pole', eps=EPS, **out)
            print('%-14s %3d steps  |x|max %.3f  |u|max %.3f  u[0] %s' % ('kalman_cart_pole', 40, np.abs(out['x']).max(), np.abs(out['u']).max(), out['u'][0]))
        if not sys.argv[1:] or 'kalman_cart_pole_np200' in sys.argv[1:]:
            # the example's OWN configuration (example_inverted_pendulum_kalman.py:19,71-110): Ts = 5 ms, Np = Nc = 200, eps_feas = 1e3, estimator
            # Q_kal = 10 I, R_kal = I; measurement noise at the level the example has commented out (0.005), linear plant, 30 steps
            import scipy
            if not hasattr(scipy, 'size'):
                scipy.size = np.size
            from pyMPC.kalman import LinearStateEstimator as RefEstimator
            kw = dict(fixtures.NAMED['cart_pole_kalman']())
            kw.update(eps_abs=EPS, eps_rel=EPS)
            out = run_kalman(RefController, RefEstimator, kw, 30, Q_kal=10 * np.eye(4), R_kal=np.eye(2), vstd=0.005, wstd=0.0)
            np.savez_compressed(os.path.join(HERE, 'traj_kalman_cart_pole_np200.npz'), pattern='output_feedback', fixture='cart_pole_kalman', eps=EPS, **out)
            print('%-14s %3d steps  |x|max %.3f  |u|max %.3f  u[0] %s' % ('kalman_cart_pole_np200', 30, np.abs(out['x']).max(), np.abs(out['u']).max(), out['u'][0]))
        for name, (fix, steps, pattern) in CASES.items():
            if sys.argv[1:] and name not in sys.argv[1:]:
                continue
            kw = dict(fixtures.NAMED[fix]())
            kw.update(eps_abs=EPS, eps_rel=EPS)
            Ad, Bd = kw['Ad'], kw['Bd']
            plant = cart_pole_plant if name == 'cart_pole' else (lambda x, u: Ad @ x + Bd @ u)
            xs, us = run(RefController, kw, steps, pattern, plant)
            np.savez_compressed(os.path.join(HERE, 'traj_%s.npz' % name), x=xs, u=us, pattern=pattern, fixture=fix, eps=EPS)
            print('%-14s %3d steps  |x|max %.3f  |u|max %.3f  u[0] %s' % (name, steps, np.abs(xs).max(), np.abs(us).max(), us[0]))


if __name__ == '__main__':
    main()
