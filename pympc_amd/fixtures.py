"""Named MPC problem instances used by the golden-vector generator, the tests and bench.py.

Every function returns a plain dict of keyword arguments for ``MPCController(**kw)``
(dense float64 ndarrays only, so the dict can be stored in an ``.npz``).

The physical constants of the named systems are *data* taken from the reference's
example scripts (cited per function, paths relative to /root/reference); the random
LTI generator is the seed-pinned recipe of SURVEY.md section 8(d).
"""
import numpy as np


def _diag(v):
    return np.diag(np.asarray(v, dtype=float))


def point_mass(Np=20):
    """examples/example_point_mass.py:11-71 (Ts=0.2, M=2, b=0.3, forward Euler)."""
    Ts, M, b = 0.2, 2.0, 0.3
    Ac = np.array([[0.0, 1.0], [0.0, -b / M]])
    Bc = np.array([[0.0], [1.0 / M]])
    return dict(
        Ad=np.eye(2) + Ac * Ts, Bd=Bc * Ts, Np=Np,
        x0=np.array([0.1, 0.2]), xref=np.array([7.0, 0.0]), uref=np.array([0.0]),
        uminus1=np.array([0.0]),
        Qx=_diag([0.5, 0.1]), QxN=_diag([0.5, 0.1]), Qu=2.0 * np.eye(1), QDu=10.0 * np.eye(1),
        xmin=np.array([-100.0, -100.0]), xmax=np.array([100.0, 100.0]),
        umin=np.array([-1.2]), umax=np.array([1.2]),
        Dumin=np.array([-0.2]), Dumax=np.array([0.2]),
    )


def cart_pole(Np=20):
    """examples/example_inverted_pendulum.py:10-69 (Ts=50 ms, eps_feas=1e3)."""
    M, m, b, ftheta, l, g, Ts = 0.5, 0.2, 0.1, 0.1, 0.3, 9.81, 50e-3
    Ac = np.array([[0, 1, 0, 0],
                   [0, -b / M, -(g * m) / M, (ftheta * m) / M],
                   [0, 0, 0, 1],
                   [0, b / (M * l), (M * g + g * m) / (M * l), -(M * ftheta + ftheta * m) / (M * l)]])
    Bc = np.array([[0.0], [1.0 / M], [0.0], [-1 / (M * l)]])
    return dict(
        Ad=np.eye(4) + Ac * Ts, Bd=Bc * Ts, Np=Np,
        x0=np.array([0.0, 0.0, 15 * 2 * np.pi / 360, 0.0]),
        xref=np.array([0.3, 0.0, 0.0, 0.0]), uref=np.array([0.0]), uminus1=np.array([0.0]),
        Qx=_diag([0.3, 0, 1.0, 0]), QxN=_diag([0.3, 0, 1.0, 0]),
        Qu=0.0 * np.eye(1), QDu=0.01 * np.eye(1),
        xmin=np.array([-1.0, -100.0, -100.0, -100.0]), xmax=np.array([0.3, 100.0, 100.0, 100.0]),
        umin=np.array([-20.0]), umax=np.array([20.0]),
        Dumin=np.array([-5.0]), Dumax=np.array([5.0]),
        eps_feas=1e3,
    )


def cart_pole_kalman(Np=200):
    """examples/example_inverted_pendulum_kalman.py:11-110: the same cart pole sampled at Ts = 5 ms, position box [-1, 1], Np = Nc = 200,
    eps_feas = 1e3 -- the reference's long-horizon output-feedback example (estimator: Q_kal = 10 I, R_kal = I, :100-103)."""
    kw = cart_pole(Np)
    M, m, b, ftheta, l, g, Ts = 0.5, 0.2, 0.1, 0.1, 0.3, 9.81, 5e-3
    Ac = np.array([[0, 1, 0, 0],
                   [0, -b / M, -(g * m) / M, (ftheta * m) / M],
                   [0, 0, 0, 1],
                   [0, b / (M * l), (M * g + g * m) / (M * l), -(M * ftheta + ftheta * m) / (M * l)]])
    Bc = np.array([[0.0], [1.0 / M], [0.0], [-1 / (M * l)]])
    kw.update(Ad=np.eye(4) + Ac * Ts, Bd=Bc * Ts, xmax=np.array([1.0, 100.0, 100.0, 100.0]))
    return kw


def accel_brake(Np=20):
    """examples/example_accelerate_brake.py:10-72 (nu=2, infinite Delta-u bounds)."""
    Ts, M, b = 0.2, 2.0, 0.3
    Ac = np.array([[0.0, 1.0], [0.0, -b / M]])
    Bc = np.array([[0.0], [1.0 / M]])
    Bc = np.c_[Bc, Bc]
    return dict(
        Ad=np.eye(2) + Ac * Ts, Bd=Bc * Ts, Np=Np,
        x0=np.array([0.1, 0.2]), xref=np.array([7.0, 0.0]), uminus1=np.array([0.0, 0.0]),
        Qx=_diag([10.0, 0.0]), QxN=_diag([10.0, 0.0]), Qu=_diag([0.5, 0.2]), QDu=_diag([0.5, 0.2]),
        xmin=np.array([-100.0, -0.8]), xmax=np.array([100.0, 0.8]),
        umin=np.array([0.0, -1.0]), umax=np.array([0.5, 0.0]),
        Dumin=np.array([-np.inf, -np.inf]), Dumax=np.array([np.inf, np.inf]),
    )


def quadcopter(Np=10, finite_du=True):
    """Quadcopter model and weights of test_scripts/main.py:7-49 (nx=12, nu=4, +-inf state bounds)."""
    Ad = np.array([
        [1., 0., 0., 0., 0., 0., 0.1, 0., 0., 0., 0., 0.],
        [0., 1., 0., 0., 0., 0., 0., 0.1, 0., 0., 0., 0.],
        [0., 0., 1., 0., 0., 0., 0., 0., 0.1, 0., 0., 0.],
        [0.0488, 0., 0., 1., 0., 0., 0.0016, 0., 0., 0.0992, 0., 0.],
        [0., -0.0488, 0., 0., 1., 0., 0., -0.0016, 0., 0., 0.0992, 0.],
        [0., 0., 0., 0., 0., 1., 0., 0., 0., 0., 0., 0.0992],
        [0., 0., 0., 0., 0., 0., 1., 0., 0., 0., 0., 0.],
        [0., 0., 0., 0., 0., 0., 0., 1., 0., 0., 0., 0.],
        [0., 0., 0., 0., 0., 0., 0., 0., 1., 0., 0., 0.],
        [0.9734, 0., 0., 0., 0., 0., 0.0488, 0., 0., 0.9846, 0., 0.],
        [0., -0.9734, 0., 0., 0., 0., 0., -0.0488, 0., 0., 0.9846, 0.],
        [0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0.9846]])
    Bd = np.array([
        [0., -0.0726, 0., 0.0726],
        [-0.0726, 0., 0.0726, 0.],
        [-0.0152, 0.0152, -0.0152, 0.0152],
        [-0., -0.0006, -0., 0.0006],
        [0.0006, 0., -0.0006, 0.0000],
        [0.0106, 0.0106, 0.0106, 0.0106],
        [0, -1.4512, 0., 1.4512],
        [-1.4512, 0., 1.4512, 0.],
        [-0.3049, 0.3049, -0.3049, 0.3049],
        [-0., -0.0236, 0., 0.0236],
        [0.0236, 0., -0.0236, 0.],
        [0.2107, 0.2107, 0.2107, 0.2107]])
    u0 = 10.5916
    inf = np.inf
    Q = _diag([0., 0., 10., 10., 10., 10., 0., 0., 0., 5., 5., 5.])
    kw = dict(
        Ad=Ad, Bd=Bd, Np=Np,
        x0=np.zeros(12), xref=np.array([0., 0., 1.] + [0.] * 9), uref=np.zeros(4), uminus1=np.zeros(4),
        Qx=Q, QxN=Q.copy(), Qu=0.1 * np.eye(4), QDu=0.0 * np.eye(4),
        xmin=np.array([-np.pi / 6, -np.pi / 6, -inf, -inf, -inf, -1.] + [-inf] * 6),
        xmax=np.array([np.pi / 6, np.pi / 6] + [inf] * 10),
        umin=np.array([9.6] * 4) - u0, umax=np.array([13.] * 4) - u0,
    )
    if finite_du:
        kw.update(Dumin=-0.5 * np.ones(4), Dumax=0.5 * np.ones(4), QDu=0.1 * np.eye(4))
    return kw


def random_lti(index, nx=12, nu=4, Np=30, xbox=10.0, ubox=1.0, dubox=0.5, eps_feas=1e6):
    """Seed-pinned random stable LTI instance.

    ``rng = default_rng(1000 + index)``; ``Ad = G * 0.95 / rho(G)``, ``G ~ N(0,1)``;
    ``Bd ~ N(0,1)``; unit state weight, 0.1 input and input-rate weights; box bounds.
    """
    rng = np.random.default_rng(1000 + int(index))
    G = rng.standard_normal((nx, nx))
    Ad = G * (0.95 / np.max(np.abs(np.linalg.eigvals(G))))
    Bd = rng.standard_normal((nx, nu))
    x0 = rng.standard_normal(nx)
    return dict(
        Ad=Ad, Bd=Bd, Np=Np,
        x0=x0, xref=np.zeros(nx), uref=np.zeros(nu), uminus1=np.zeros(nu),
        Qx=np.eye(nx), QxN=np.eye(nx), Qu=0.1 * np.eye(nu), QDu=0.1 * np.eye(nu),
        xmin=-xbox * np.ones(nx), xmax=xbox * np.ones(nx),
        umin=-ubox * np.ones(nu), umax=ubox * np.ones(nu),
        Dumin=-dubox * np.ones(nu), Dumax=dubox * np.ones(nu),
        eps_feas=eps_feas,
    )


def random_lti_noise_rng(index):
    """Process-noise stream for the receding-horizon simulation of instance ``index``
    (kept separate from the model stream so model data do not depend on the step count)."""
    return np.random.default_rng(500000 + int(index))


def small_mimo(Np=3):
    """Tiny nx=2, nu=2 system exposing the Delta-u row structure of mpc.py:569-580 for nu>1."""
    return dict(
        Ad=np.array([[1.0, 0.1], [-0.2, 0.9]]), Bd=np.array([[0.0, 0.1], [0.2, -0.1]]), Np=Np,
        x0=np.array([0.5, -0.3]), xref=np.array([1.0, 0.0]), uref=np.array([0.1, -0.1]),
        uminus1=np.array([0.05, 0.02]),
        Qx=np.array([[2.0, 0.3], [0.3, 1.0]]), QxN=np.array([[4.0, 0.1], [0.1, 3.0]]),
        Qu=np.array([[0.5, 0.1], [0.1, 0.4]]), QDu=np.array([[1.0, 0.2], [0.2, 0.7]]),
        xmin=np.array([-2.0, -1.0]), xmax=np.array([2.0, 1.0]),
        umin=np.array([-1.0, -0.5]), umax=np.array([1.0, 0.5]),
        Dumin=np.array([-0.3, -0.2]), Dumax=np.array([0.3, 0.25]),
        eps_feas=1e4,
    )


def point_mass_nc(Np=25, Nc=10):
    """Control horizon Nc < Np with a 2-D (Np+1, nx) reference: the __main__ demo of pyMPC/mpc.py:618-676."""
    kw = point_mass(Np)
    kw.update(Nc=Nc, xmin=np.array([-10.0, -10.0]), xmax=np.array([7.0, 10.0]))
    kw['xref'] = np.kron(np.ones((Np + 1, 1)), np.array([7.0, 0.0]))
    return kw


NAMED = {
    'point_mass': point_mass,
    'cart_pole': cart_pole,
    'cart_pole_kalman': cart_pole_kalman,         # the reference's long-horizon output-feedback example: (4, 1, Np = Nc = 200)
    'accel_brake': accel_brake,
    'quadcopter': quadcopter,
    'quadcopter_nodu': lambda: quadcopter(finite_du=False),
    'small_mimo': small_mimo,
    'point_mass_nc': point_mass_nc,
    'random_12_4_30': lambda: random_lti(0),
    'random_12_4_30_b': lambda: random_lti(7),
    'random_20_8_12': lambda: random_lti(3, nx=20, nu=8, Np=12, xbox=1.0),
    'random_20_8_100': lambda: random_lti(0, nx=20, nu=8, Np=100, xbox=1.0),      # BASELINE cfg-5, instance 0 of bench.py --workload cfg5
    'random_5_3_8': lambda: random_lti(11, nx=5, nu=3, Np=8, xbox=0.5),
    'random_5_3_8_nc': lambda: dict(random_lti(12, nx=5, nu=3, Np=8, xbox=2.0), Nc=3),
    'quadcopter_nc': lambda: dict(quadcopter(Np=10), Nc=4),
    'cart_pole_nc1': lambda: dict(cart_pole(Np=12), Nc=1),
    # SOFT_ON = False (hidden switch, mpc.py:237): hard state box, no slack columns.  Keys starting with '_' are attributes set
    # on the controller after construction, not constructor arguments.
    'point_mass_hard': lambda: dict(point_mass(), _SOFT_ON=False),
    'accel_brake_hard': lambda: dict(accel_brake(), _SOFT_ON=False),              # the velocity bound 0.8 is active along the horizon
    'random_12_4_30_hard': lambda: dict(random_lti(7), _SOFT_ON=False),
    'random_5_3_8_nc_hard': lambda: dict(random_lti(12, nx=5, nu=3, Np=8, xbox=2.0), Nc=3, _SOFT_ON=False),
    'random_20_8_12_hard': lambda: dict(random_lti(3, nx=20, nu=8, Np=12, xbox=3.0), _SOFT_ON=False),
}


def split_attrs(kw):
    """(constructor kwargs, attributes to set afterwards) of a fixture dict."""
    return ({k: v for k, v in kw.items() if not k.startswith('_')}, {k[1:]: v for k, v in kw.items() if k.startswith('_')})
