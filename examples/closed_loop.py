"""The reference's example loops as one driver on the drop-in class:

    K = MPCController(Ad, Bd, Np=..., x0=..., ...);  K.setup()
    for k in range(nsim):  u = K.output();  x = plant(x, u);  K.update(x)            # examples/example_point_mass.py:88-101

for the systems of examples/example_point_mass.py, example_inverted_pendulum.py (nonlinear cart pole under the linear MPC), example_accelerate_brake.py and
the quadcopter of test_scripts/main.py (constants: pympc_amd/fixtures.py, cited there).  Printed: the latency of update() -- the reference prints its own, ~1 ms per
step on a laptop CPU --, the end state, and, run at a tight tolerance (--eps 1e-10) where tests/golden/ holds the reference classes' own trajectory of the
system, the largest distance to it.
--device-loop B: the same loop for B copies of the controller INSIDE one kernel launch (BatchMPCController.run, linear plant = the model: mpcqp_mpc_loop).

    python examples/closed_loop.py cart_pole [--steps 120] [--eps 1e-3] [--device-loop 1024]
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


def cart_pole_plant(x, u, Ts=50e-3):
    """the nonlinear pendulum on a cart, one forward-Euler step (example_inverted_pendulum.py:10-17,92-103)"""
    M, m, b, ft, l, g = 0.5, 0.2, 0.1, 0.1, 0.3, 9.81
    F, v, th, om = float(u[0]), x[1], x[2], x[3]
    s, c = np.sin(th), np.cos(th)
    den = M + m * (1.0 - c * c)
    acc = (m * l * s * om ** 2 - m * g * s * c + m * ft * c * om + F - b * v) / den
    alp = ((M + m) * (g * s - ft * om) - m * l * om ** 2 * s * c - (F - b * v) * c) / (l * den)
    return x + Ts * np.array([v, acc, om, alp])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('system', choices=['point_mass', 'cart_pole', 'accel_brake', 'quadcopter'])
    ap.add_argument('--steps', type=int, default=None); ap.add_argument('--eps', type=float, default=1e-3)
    ap.add_argument('--device-loop', type=int, default=0, metavar='B', help='also run B copies of the loop inside one kernel launch (linear plant)')
    a = ap.parse_args()
    kw, attrs = fixtures.split_attrs(dict(fixtures.NAMED[a.system](), eps_abs=a.eps, eps_rel=a.eps))
    Ad, Bd = np.asarray(kw['Ad'], dtype=float), np.asarray(kw['Bd'], dtype=float).reshape(np.asarray(kw['Ad']).shape[0], -1)
    plant = cart_pole_plant if a.system == 'cart_pole' else (lambda x, u: Ad @ x + Bd @ np.atleast_1d(u))
    gpath = os.path.join(ROOT, 'tests', 'golden', 'traj_%s.npz' % a.system)
    gold = np.load(gpath, allow_pickle=True) if os.path.exists(gpath) else None
    nsim = a.steps or (len(gold['u']) if gold is not None else 100)
    K = MPCController(**kw)
    for k_, v_ in attrs.items():
        setattr(K, k_, v_)
    if a.eps < 1e-6:
        K.solver_settings = dict(max_iter=400000)              # (OSQP's default of 4000 iterations is for its default tolerance)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        t0 = time.perf_counter(); K.setup(); t_setup = time.perf_counter() - t0
        x, u = np.array(kw['x0'], dtype=float), np.array(kw['uminus1'], dtype=float)
        xs, us, lat, bad = [x], [], [], 0
        order = str(gold['pattern']) if gold is not None else 'output_update'      # (the reference's examples call update() before or after output(); the goldens say which)
        for k in range(nsim):
            if order == 'update_output':
                t0 = time.perf_counter(); K.update(x, u); lat.append(time.perf_counter() - t0)
                u = K.output()
            else:
                u = K.output()
            x = plant(x, u)
            if order != 'update_output':
                t0 = time.perf_counter(); K.update(x); lat.append(time.perf_counter() - t0)
            bad += K.res.info.status != 'solved'
            xs.append(x); us.append(np.atleast_1d(u).copy())
    xs, us, lat = np.array(xs), np.array(us), 1e6 * np.array(lat)
    print('%s: (nx, nu, Np) = (%d, %d, %d), %d steps, eps %.0e; setup() %.1f ms (the first of the process: it loads the code objects of the library); update() median %.1f us, p95 %.1f us; %d solves not "solved"'
          % (a.system, Ad.shape[0], Bd.shape[1], kw['Np'], nsim, a.eps, 1e3 * t_setup, np.median(lat), np.percentile(lat, 95), bad))
    print('  end state', np.round(xs[-1], 5), ' last input', np.round(us[-1], 5))
    if gold is not None and a.steps is None and a.eps <= 1e-9:
        print('  largest distance to the reference classes\' own trajectory (tests/golden/traj_%s.npz, made at eps %.0e): x %.2e, u %.2e'
              % (a.system, float(gold['eps']), np.abs(xs - gold['x']).max(), np.abs(us - gold['u']).max()))
    if a.device_loop:
        B = a.device_loop
        st = lambda v: np.broadcast_to(np.asarray(v, dtype=float), (B,) + np.shape(v))
        rng = np.random.default_rng(0)
        X0 = np.asarray(kw['x0'], dtype=float)[None, :] * rng.uniform(0.5, 1.0, size=(B, 1))
        names = ('xref', 'uref', 'uminus1', 'Qx', 'QxN', 'Qu', 'QDu', 'xmin', 'xmax', 'umin', 'umax', 'Dumin', 'Dumax')
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            Kb = BatchMPCController(st(Ad), st(Bd), Np=kw['Np'], Nc=kw.get('Nc'), x0=X0, eps_feas=np.full((B, 1), kw.get('eps_feas', 1e6)),
                                    eps_abs=a.eps, eps_rel=a.eps, **{k: st(kw[k]) for k in names if k in kw})
            for k_, v_ in attrs.items():
                setattr(Kb, k_, v_)
            if a.eps < 1e-6:
                Kb.solver_settings = dict(max_iter=400000)
            Kb.setup()
            Kb.run(min(nsim, 5))                                   # (first use: the code object of the loop kernel loads)
            t0 = time.perf_counter(); tr = Kb.run(nsim); t_loop = time.perf_counter() - t0
        print('  device loop: %d controllers x %d steps in %.2f ms (%.0f MPC steps per second), %d of %d solves "solved"; kernel %s'
              % (B, nsim, 1e3 * t_loop, B * nsim / t_loop, int((tr['status'] == 1).sum()), B * nsim, Kb.prob.kernel_name(True)))


if __name__ == '__main__':
    main()
