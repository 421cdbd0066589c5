"""Seeded random controllers over the whole range of shapes the device takes -- 1 <= nx + nu <= 64 (all four KKT backends), Nc <= Np, soft and
hard state constraints (pyMPC's SOFT_ON, mpc.py:237), batches of 1..4 -- (a) cold solve and one warm step against the CPU oracle at tight
tolerance, (b) the closed loop on the device (mpcqp_mpc_loop) against output() / update() per step, bit for bit.  A committed slice of
scripts/fuzz_parity.py and scripts/fuzz_loop.py (620 + 150 cases there)."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _shape(rng):
    kind = int(rng.integers(0, 4))
    if kind == 0: nx, nu, Np = int(rng.integers(1, 5)), int(rng.integers(1, 3)), int(rng.integers(2, 12))
    elif kind == 1: nx, nu, Np = int(rng.integers(4, 13)), int(rng.integers(1, 5)), int(rng.integers(8, 40))
    elif kind == 2: nx, nu, Np = int(rng.integers(14, 25)), int(rng.integers(2, 9)), int(rng.integers(3, 30))
    else: nx, nu, Np = int(rng.integers(26, 50)), int(rng.integers(2, 12)), int(rng.integers(3, 10))
    if nx + nu > 64:                      # (64 < nx + nu <= 128 has its own cases in tests/test_gpu_wide.py; the seeds here keep their shapes)
        nx = 64 - nu
    Nc = int(rng.integers(1, Np + 1)) if rng.random() < 0.4 else Np
    return nx, nu, Np, Nc


@pytest.mark.parametrize('seed', range(32))
def test_random_shape_matches_oracle(seed):
    from pympc_amd import MPCController, fixtures
    from oracle.osqp_oracle import OSQP
    rng = np.random.default_rng(61000 + seed)
    nx, nu, Np, Nc = _shape(rng)
    kw = dict(fixtures.random_lti(62000 + seed, nx=nx, nu=nu, Np=Np, xbox=4.0))
    kw['x0'] = 0.4 * kw['x0']
    kw.update(Nc=Nc, eps_abs=1e-10, eps_rel=1e-10)
    K = MPCController(**kw); Ko = MPCController(**kw); Ko.prob = OSQP()
    K.solver_settings = Ko.solver_settings = dict(max_iter=400000)
    K.SOFT_ON = Ko.SOFT_ON = bool(rng.random() < 0.75)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        K.setup(); Ko.setup()
        assert K.res.info.status == Ko.res.info.status
        if Ko.res.info.status != 'solved':
            return
        (u, info), (uo, infoo) = K.output(return_u_seq=True), Ko.output(return_u_seq=True)
        scale = max(1e-3, np.abs(infoo['u_seq']).max())
        assert np.abs(info['u_seq'] - infoo['u_seq']).max() <= 1e-6 * scale          # (measured over 620 such cases: <= 4e-12)
        x = kw['Ad'] @ kw['x0'] + kw['Bd'] @ uo
        K.update(x, uo); Ko.update(x, uo)
        assert K.res.info.status == Ko.res.info.status
        assert np.abs(K.output() - Ko.output()).max() <= 1e-6 * scale


@pytest.mark.parametrize('seed', range(24))
def test_random_shape_device_loop_equals_stepwise(seed):
    from pympc_amd import BatchMPCController, fixtures
    rng = np.random.default_rng(63000 + seed)
    nx, nu, Np, Nc = _shape(rng)
    B = int(rng.integers(1, 5))
    soft = bool(rng.random() < 0.75)
    kws = [fixtures.random_lti(64000 + 7 * seed + i, nx=nx, nu=nu, Np=Np, xbox=4.0) for i in range(B)]
    keys = ('x0', 'xref', 'uref', 'uminus1', 'Qx', 'QxN', 'Qu', 'QDu', 'xmin', 'xmax', 'umin', 'umax', 'Dumin', 'Dumax')
    stack = lambda k: np.stack([np.asarray(kw[k], dtype=float) for kw in kws])

    def make():
        K = BatchMPCController(stack('Ad'), stack('Bd'), Np=Np, Nc=Nc, eps_feas=np.array([[kw.get('eps_feas', 1e6)] for kw in kws]),
                               SOFT_ON=soft, **{k: stack(k) for k in keys})
        K.setup()
        return K
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Kd, Ks = make(), make()
        steps = 5
        w = 0.01 * rng.standard_normal((steps, B, nx))
        tr = Kd.run(steps, w=w)
        assert np.array_equal(tr['x'][0], Ks.x0_rh)
        for k in range(steps):
            assert np.array_equal(Ks.output(), tr['u'][k]), k
            Ks.update(tr['x'][k + 1])
            infos = Ks.prob.infos()
            assert [i.status for i in infos] == list(tr['status'][k]) and [i.iter for i in infos] == list(tr['iter'][k]), k
