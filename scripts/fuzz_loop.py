"""One-off sweep: the closed loop on the device (mpcqp_mpc_loop) against the stepwise API (output() / update() per step) for seeded random
shapes -- all four KKT backends, Nc < Np, soft and hard state constraints, batches of 1..5 controllers.  The two must agree bit for bit
(same kernels, same data; tests/test_gpu_loop_parity.py does this for four fixtures).   python scripts/fuzz_loop.py [first_seed] [count]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pympc_amd import BatchMPCController, fixtures

first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 80)
keys = ('x0', 'xref', 'uref', 'uminus1', 'Qx', 'QxN', 'Qu', 'QDu', 'xmin', 'xmax', 'umin', 'umax', 'Dumin', 'Dumax')
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(52000 + seed)
    kind = rng.integers(0, 4)
    if kind == 0: nx, nu, Np = int(rng.integers(1, 5)), int(rng.integers(1, 3)), int(rng.integers(2, 12))          # small: dense backend if it fits
    elif kind == 1: nx, nu, Np = int(rng.integers(4, 13)), int(rng.integers(1, 5)), int(rng.integers(8, 40))        # 16-wide stages
    elif kind == 2: nx, nu, Np = int(rng.integers(14, 25)), int(rng.integers(2, 9)), int(rng.integers(3, 30))       # 32-wide
    else: nx, nu, Np = int(rng.integers(26, 50)), int(rng.integers(2, 12)), int(rng.integers(3, 10))                # 64-wide
    if nx + nu > 64: nx = 64 - nu
    Nc = int(rng.integers(1, Np + 1)) if rng.random() < 0.4 else Np
    B = int(rng.integers(1, 6))
    soft = bool(rng.random() < 0.8)
    kws = [fixtures.random_lti(53000 + 7 * seed + i, nx=nx, nu=nu, Np=Np, xbox=4.0) for i in range(B)]
    stack = lambda k: np.stack([np.asarray(kw[k], dtype=float) for kw in kws])

    def make():
        K = BatchMPCController(stack('Ad'), stack('Bd'), Np=Np, Nc=Nc, eps_feas=np.array([[kw.get('eps_feas', 1e6)] for kw in kws]),
                               SOFT_ON=soft, **{k: stack(k) for k in keys})
        K.setup()
        return K
    tag = 'seed %d nx=%d nu=%d Np=%d Nc=%d B=%d soft=%d' % (seed, nx, nu, Np, Nc, B, soft)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            Kd, Ks = make(), make()
            steps = 6
            w = 0.01 * rng.standard_normal((steps, B, nx))
            tr = Kd.run(steps, w=w)
            ok = np.array_equal(tr['x'][0], Ks.x0_rh)
            for k in range(steps):
                u = Ks.output()
                ok = ok and np.array_equal(u, tr['u'][k])
                Ks.update(tr['x'][k + 1])
                infos = Ks.prob.infos()
                ok = ok and [i.status for i in infos] == list(tr['status'][k]) and [i.iter for i in infos] == list(tr['iter'][k])
            msg = bp = Kd.prob.kernel_name(loop=True)
    except Exception as e:                                   # noqa: BLE001
        ok, msg = False, 'exception %r' % (e,)
    if not ok:
        bad += 1
    print(('ok   ' if ok else 'FAIL ') + tag + '  ' + msg, flush=True)
print('%d of %d failed' % (bad, count))
