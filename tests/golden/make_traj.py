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

    python tests/golden/make_traj.py        # needs /root/reference; writes tests/golden/traj_<name>.npz
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
class OSQP(_Oracle):
    def setup(self, *a, **kw):
        kw.setdefault('max_iter', 400000)
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


def main():
    from pympc_amd import fixtures
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, 'osqp'))
        with open(os.path.join(tmp, 'osqp', '__init__.py'), 'w') as f:
            f.write(STUB)
        sys.path.insert(0, tmp)
        sys.path.insert(0, '/root/reference')
        from pyMPC.mpc import MPCController as RefController
        for name, (fix, steps, pattern) in CASES.items():
            kw = dict(fixtures.NAMED[fix]())
            kw.update(eps_abs=EPS, eps_rel=EPS)
            Ad, Bd = kw['Ad'], kw['Bd']
            plant = cart_pole_plant if name == 'cart_pole' else (lambda x, u: Ad @ x + Bd @ u)
            xs, us = run(RefController, kw, steps, pattern, plant)
            np.savez_compressed(os.path.join(HERE, 'traj_%s.npz' % name), x=xs, u=us, pattern=pattern, fixture=fix, eps=EPS)
            print('%-14s %3d steps  |x|max %.3f  |u|max %.3f  u[0] %s' % (name, steps, np.abs(xs).max(), np.abs(us).max(), us[0]))


if __name__ == '__main__':
    main()
