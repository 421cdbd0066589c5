"""One-off sweep: B copies of ONE seeded random controller, states scattered, closed loop on the device -- on ONE shared KKT factor (what every setup arranges,
mpcqp_share_factor) against every instance on its own (MPCQP_TUNE_NO_SHARE).  Inputs, states, statuses, iteration counts and the solver counters must agree
bit for bit, across rho updates that make instances leave the shared slot in mid-launch (tight tolerance, tight state box: they do happen).
    python scripts/fuzz_share.py [first_seed] [count]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pympc_amd import BatchMPCController, fixtures, _lib

first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 60)
keys = ('xref', 'uref', 'uminus1', 'Qx', 'QxN', 'Qu', 'QDu', 'xmin', 'xmax', 'umin', 'umax', 'Dumin', 'Dumax')
bad = refs = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(62000 + seed)
    kind = rng.integers(0, 4)
    if kind == 0: nx, nu, Np = int(rng.integers(1, 5)), int(rng.integers(1, 3)), int(rng.integers(30, 70))          # small stages, long horizon: grouped stages
    elif kind == 1: nx, nu, Np = int(rng.integers(4, 13)), int(rng.integers(1, 5)), int(rng.integers(8, 40))        # 16-wide stages
    elif kind == 2: nx, nu, Np = int(rng.integers(14, 25)), int(rng.integers(2, 9)), int(rng.integers(3, 30))       # 32-wide
    else: nx, nu, Np = int(rng.integers(26, 50)), int(rng.integers(2, 12)), int(rng.integers(3, 10))                # 64-wide
    if nx + nu > 64: nx = 64 - nu
    Nc = int(rng.integers(1, Np + 1)) if rng.random() < 0.4 else Np
    B = int(rng.integers(2, 40))
    soft = bool(rng.random() < 0.8)
    eps = float(10.0 ** -rng.integers(3, 9))
    kw = fixtures.random_lti(63000 + seed, nx=nx, nu=nu, Np=Np, xbox=float(rng.choice([1.0, 4.0, 10.0])))
    st = lambda a: np.broadcast_to(np.asarray(a, dtype=float), (B,) + np.shape(a))
    X0 = kw['x0'][None, :] * rng.uniform(0.2, 2.5, size=(B, 1)) * rng.choice([-1.0, 1.0], size=(B, nx))
    steps = 6
    w = 0.01 * rng.standard_normal((steps, B, nx))

    def walk(tuning):
        K = BatchMPCController(st(kw['Ad']), st(kw['Bd']), Np=Np, Nc=Nc, x0=st(kw['x0']), eps_feas=np.full((B, 1), kw.get('eps_feas', 1e6)), SOFT_ON=soft,
                               eps_abs=eps, eps_rel=eps, **{k: st(kw[k]) for k in keys})
        K.solver_settings = dict(backend='sweeps', tuning=tuning, max_iter=20000)
        K.setup()
        sharing = K.share_factor()
        K.update(X0)
        u0 = K.output().copy()
        tr = K.run(steps, w=w)
        return sharing, u0, tr, K.prob.stats(), K.prob.kernel_name(loop=True)
    tag = 'seed %d nx=%d nu=%d Np=%d Nc=%d B=%d soft=%d eps=%.0e' % (seed, nx, nu, Np, Nc, B, soft, eps)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            sh, u_s, tr_s, st_s, kn = walk(0)
            _, u_o, tr_o, st_o, _ = walk(_lib.TUNE_NO_SHARE)
        ok = sh == B and np.array_equal(u_s, u_o) and st_s == st_o and all(np.array_equal(tr_s[k], tr_o[k]) for k in ('x', 'u', 'status', 'iter'))
        refs += st_s[2]
        msg = '%s  refactorizations %d' % (kn, st_s[2])
    except Exception as e:                                   # noqa: BLE001
        ok, msg = False, 'exception %r' % (e,)
    if not ok:
        bad += 1
    print(('ok   ' if ok else 'FAIL ') + tag + '  ' + msg, flush=True)
print('%d of %d failed; %d refactorizations (instances leaving the shared factor) on the way' % (bad, count, refs))
