"""The reference's Monte-Carlo use of ONE controller as a function (test_scripts/example_mpc_function.py:105-111):

    for i in range(N_mc):                                  # 10 000 random (x, u_{-1})
        uMPC = K.__controller_function__(x_i, uminus1_i)

-- 10 000 sequential update() + solve() calls there.  The states are independent, so the batch surface evaluates the same map in ONE call:
every (x_i, u_{-1,i}) becomes an instance of a BatchMPCController that broadcasts the model (point mass, nx = 2, nu = 1, Np = 20; the constants
of the reference script), set up and solved in a single launch.  Printed: the time of both ways and the largest difference of the two results
(each is an ADMM iterate at the solver tolerance; run with a tight tolerance, `--eps 1e-9`, to see them agree to 1e-7).

    python examples/controller_map_monte_carlo.py [--n 10000] [--eps 1e-3] [--sequential 500]
"""
import argparse
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pympc_amd import BatchMPCController, MPCController            # noqa: E402


def point_mass(eps):
    Ts, M, b = 0.2, 2.0, 0.3                                        # example_mpc_function.py:11-13
    Ad = np.array([[1.0, Ts], [0.0, 1.0 - b / M * Ts]])
    Bd = np.array([[0.0], [Ts / M]])
    return dict(Ad=Ad, Bd=Bd, Np=20, xref=np.array([7.0, 0.0]), uref=np.array([0.0]),
                Qx=np.diag([0.5, 0.1]), QxN=np.diag([0.5, 0.1]), Qu=2.0 * np.eye(1), QDu=10.0 * np.eye(1),
                xmin=np.array([-100.0, -100.0]), xmax=np.array([100.0, 100.0]), umin=np.array([-1.2]), umax=np.array([1.2]),
                Dumin=np.array([-2e-1]), Dumax=np.array([2e-1]), eps_abs=eps, eps_rel=eps)


def controller_map(kw, X, Um1, max_iter=None):
    """u_i = K(x_i, u_{-1,i}) for all rows of X, Um1 at once: one batched setup (cold start) + solve."""
    B = X.shape[0]
    stack = lambda a: np.broadcast_to(np.asarray(a, dtype=float), (B,) + np.shape(a))
    extra = dict(max_iter=max_iter) if max_iter else {}
    K = BatchMPCController(stack(kw['Ad']), stack(kw['Bd']), Np=kw['Np'], x0=X, uminus1=Um1, xref=stack(kw['xref']), uref=stack(kw['uref']),
                           Qx=stack(kw['Qx']), QxN=stack(kw['QxN']), Qu=stack(kw['Qu']), QDu=stack(kw['QDu']),
                           xmin=stack(kw['xmin']), xmax=stack(kw['xmax']), umin=stack(kw['umin']), umax=stack(kw['umax']),
                           Dumin=stack(kw['Dumin']), Dumax=stack(kw['Dumax']), eps_abs=kw['eps_abs'], eps_rel=kw['eps_rel'], **extra)
    K.setup()
    return K.output(), K


def controller_map_one_controller(kw, X, Um1, x_setup, um1_setup, max_iter=None, backend=None):
    """The same map the way the reference's loop runs it: ONE controller is set up (at x_setup, um1_setup -- example_mpc_function.py:61-64), then only its
    state moves.  Here: B copies of that one controller (same model, same setup state: the batch shares ONE KKT factor, mpcqp_share_factor -- setup()
    arranges it on the streaming backends), update() scatters the states, every instance warm-starts from the setup solution."""
    B = X.shape[0]
    stack = lambda a: np.broadcast_to(np.asarray(a, dtype=float), (B,) + np.shape(a))
    K = BatchMPCController(stack(kw['Ad']), stack(kw['Bd']), Np=kw['Np'], x0=stack(x_setup), uminus1=stack(um1_setup), xref=stack(kw['xref']), uref=stack(kw['uref']),
                           Qx=stack(kw['Qx']), QxN=stack(kw['QxN']), Qu=stack(kw['Qu']), QDu=stack(kw['QDu']),
                           xmin=stack(kw['xmin']), xmax=stack(kw['xmax']), umin=stack(kw['umin']), umax=stack(kw['umax']),
                           Dumin=stack(kw['Dumin']), Dumax=stack(kw['Dumax']), eps_abs=kw['eps_abs'], eps_rel=kw['eps_rel'])
    K.solver_settings = dict(({'max_iter': max_iter} if max_iter else {}), **({'backend': backend} if backend else {}))
    K.setup()
    sharing = K.share_factor()
    K.update(X, Um1)
    return K.output(), K, sharing


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=10000); ap.add_argument('--eps', type=float, default=1e-3)
    ap.add_argument('--sequential', type=int, default=500, help='how many of the states also go through the single controller, one call each')
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    X, Um1 = rng.random((a.n, 2)), rng.random((a.n, 1))             # example_mpc_function.py:108-109
    kw = point_mass(a.eps)
    big = dict(max_iter=200000) if a.eps < 1e-6 else {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        controller_map(kw, X[:64], Um1[:64], **big)                 # (first use of the library: code objects load)
        t = time.perf_counter(); U, _ = controller_map(kw, X, Um1, **big); t_batch = time.perf_counter() - t
        K = MPCController(x0=X[0], uminus1=Um1[0], **kw)
        if big: K.solver_settings = dict(big)
        K.setup()
        m = min(a.sequential, a.n)
        t = time.perf_counter()
        Us = np.array([K.__controller_function__(X[i], Um1[i]) for i in range(m)])
        t_seq = time.perf_counter() - t
    print('batched map : %d states in %.1f ms (%.0f per second; setup of %d instances included)' % (a.n, 1e3 * t_batch, a.n / t_batch, a.n))
    print('sequential  : %d states in %.1f ms (%.0f per second; warm-started from the previous, unrelated state as in the reference loop)' % (m, 1e3 * t_seq, m / t_seq))
    print('largest |u_batched - u_sequential| over those %d: %.2e (solver tolerance %.0e)' % (m, np.abs(U[:m] - Us).max(), a.eps))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        t = time.perf_counter(); U1, _, sharing = controller_map_one_controller(kw, X, Um1, X[0], Um1[0], **big); t_one = time.perf_counter() - t
    print('one controller, many states: %d states in %.1f ms (%.0f per second; %d instances on one shared KKT factor); largest |u - u_batched| %.2e'
          % (a.n, 1e3 * t_one, a.n / t_one, sharing, np.abs(U1 - U).max()))


if __name__ == '__main__':
    main()
