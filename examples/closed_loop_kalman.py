"""The reference's output-feedback example (examples/example_inverted_pendulum_kalman.py:135-174) on the drop-in classes: the cart pole at Ts = 5 ms with a long
horizon (Np = Nc = 200), a Kalman filter (pympc_amd.kalman: kalman_design_simple, LinearStateEstimator) between the noisy measurement of position and angle and
the controller:

    y = C x + v;   u = K.output();   x = plant(x, u) + w;   KF.update(y); KF.predict(u);   K.update(KF.x, u)

Plant here: the linear model (as in the recorded run tests/golden/traj_kalman_cart_pole_np200.npz, whose noise realisation and filter gain are used so that the
distance to the reference classes' own run can be printed).  --device-loop B: B copies of the loop, estimator included, inside ONE kernel launch
(BatchMPCController.run(estimator=...): mpcqp_mpc_loop with ny > 0).

    python examples/closed_loop_kalman.py [--eps 1e-10] [--device-loop 256]
"""
import argparse
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pympc_amd import BatchMPCController, MPCController, fixtures      # noqa: E402
from pympc_amd.kalman import BatchLinearStateEstimator, LinearStateEstimator, kalman_design_simple      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--eps', type=float, default=1e-3); ap.add_argument('--device-loop', type=int, default=0, metavar='B')
    a = ap.parse_args()
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'traj_kalman_cart_pole_np200.npz'), allow_pickle=True)
    kw = dict(fixtures.cart_pole_kalman(), eps_abs=a.eps, eps_rel=a.eps)
    Ad, Bd, C = kw['Ad'], kw['Bd'], g['C']
    nx, nu, ny = 4, 1, C.shape[0]
    # the example's design: Q_kal = 10 I, R_kal = I (example_inverted_pendulum_kalman.py:100-103); the recorded run's gain is the same design
    L = kalman_design_simple(Ad, Bd, C, np.zeros((ny, nu)), 10.0 * np.eye(nx), np.eye(ny), type='filter')[0]
    print('Kalman gain: distance of this design to the recorded one %.1e' % np.abs(L - g['L']).max())
    big = dict(max_iter=400000) if a.eps < 1e-6 else {}
    K = MPCController(**kw); K.solver_settings = dict(big)
    KF = LinearStateEstimator(np.array(kw['x0'], dtype=float), Ad, Bd, C, np.zeros((ny, nu)), g['L'])
    x = np.array(g['x_true0'], dtype=float)
    nsim = len(g['u'])
    xs, us, lat = [x], [], []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup()
        for k in range(nsim):
            y = C @ x + g['v'][k]
            u = K.output()
            x = Ad @ x + Bd @ u + g['w'][k]
            KF.update(y); KF.predict(u)
            t0 = time.perf_counter(); K.update(KF.x, u); lat.append(time.perf_counter() - t0)
            xs.append(x); us.append(u.copy())
    xs, us, lat = np.array(xs), np.array(us), 1e6 * np.array(lat)
    print('cart pole, output feedback: (nx, nu, Np) = (4, 1, 200), %d steps, eps %.0e; update() median %.1f us, p95 %.1f us' % (nsim, a.eps, np.median(lat), np.percentile(lat, 95)))
    if a.eps <= 1e-9:
        print('  largest distance to the reference classes\' own run (made at eps %.0e): x %.2e, u %.2e' % (float(g['eps']), np.abs(xs - g['x']).max(), np.abs(us - g['u']).max()))
    if a.device_loop:
        B = a.device_loop
        st = lambda v: np.broadcast_to(np.asarray(v, dtype=float), (B,) + np.shape(v)).copy()
        names = ('x0', 'xref', 'uref', 'uminus1', 'Qx', 'QxN', 'Qu', 'QDu', 'xmin', 'xmax', 'umin', 'umax', 'Dumin', 'Dumax')
        rng = np.random.default_rng(0)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            Kb = BatchMPCController(st(Ad), st(Bd), Np=kw['Np'], eps_feas=np.full((B, 1), kw['eps_feas']), eps_abs=a.eps, eps_rel=a.eps, **{k: st(kw[k]) for k in names}, **big)
            Kb.setup()
            est = BatchLinearStateEstimator(st(kw['x0']), st(Ad), st(Bd), st(C), st(g['L']), x_true=st(g['x_true0']), v=0.01 * rng.standard_normal((nsim, B, ny)))
            t0 = time.perf_counter(); tr = Kb.run(nsim, w=0.001 * rng.standard_normal((nsim, B, nx)), estimator=est); t_loop = time.perf_counter() - t0
        print('  device loop with the estimator in the kernel: %d controllers x %d steps in %.1f ms (%.0f MPC steps per second), %d of %d solves "solved"; kernel %s'
              % (B, nsim, 1e3 * t_loop, B * nsim / t_loop, int((tr['status'] == 1).sum()), B * nsim, Kb.prob.kernel_name(True)))


if __name__ == '__main__':
    main()
