"""One-off sweep: seeded random controllers (tests/test_gpu_parity.py::_random_case with larger dimension ranges: 32-, 64- and 128-wide
stages, small stages on long horizons, every cyclic-reduction schedule, Nc < Np, hard and soft state constraints) -- cold solve and one warm step against the CPU oracle at tight tolerance.
python scripts/fuzz_parity.py [first_seed] [count]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_gpu_parity as T
from pympc_amd import fixtures

first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 120)
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(31000 + seed)
    kw = T._random_case(seed)
    kind = rng.random()
    if kind < 0.75:                                         # widen: the generator of the test-suite stops at nx = 13, nu = 5
        if kind < 0.45: nx, nu, Np = int(rng.integers(10, 50)), int(rng.integers(2, 14)), int(rng.integers(3, 16))
        elif kind < 0.6: nx, nu, Np = int(rng.integers(1, 7)), int(rng.integers(1, 3)), int(rng.integers(32, 130))        # small stages, long horizons (grouped: round 4)
        elif kind < 0.7: nx, nu, Np = int(rng.integers(3, 13)), int(rng.integers(1, 4)), int(rng.integers(8, 31))        # cyclic-reduction territory, every schedule
        else: nx, nu, Np = int(rng.integers(55, 110)), int(rng.integers(4, 18)), int(rng.integers(2, 5))                  # 65 .. 128 wide (round 4)
        if kind < 0.45 and nx + nu > 64:
            nx = 64 - nu
        if nx + nu > 128:
            nx = 128 - nu
        kw = dict(fixtures.random_lti(41000 + seed, nx=nx, nu=nu, Np=Np, xbox=4.0)); kw['x0'] = 0.4 * kw['x0']
        if rng.random() < 0.4:
            kw['Nc'] = int(rng.integers(1, Np + 1))
    soft = rng.random() < 0.8
    kw.update(eps_abs=1e-10, eps_rel=1e-10)
    K = T._gpu_controller(kw, max_iter=400000); Ko = T._oracle_controller(kw, max_iter=400000)
    K.SOFT_ON = Ko.SOFT_ON = soft
    tag = 'seed %d nx=%d nu=%d Np=%d Nc=%s soft=%d' % (seed, kw['Ad'].shape[0], kw['Bd'].shape[1], kw['Np'], kw.get('Nc'), soft)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            K.setup(); Ko.setup()
            ok = K.res.info.status == Ko.res.info.status
            if Ko.res.info.status == 'solved':
                (u, info), (uo, infoo) = K.output(return_u_seq=True), Ko.output(return_u_seq=True)
                scale = max(1e-3, np.abs(infoo['u_seq']).max())
                err = np.abs(info['u_seq'] - infoo['u_seq']).max() / scale
                x = kw['Ad'] @ kw['x0'] + kw['Bd'] @ uo
                K.update(x, uo); Ko.update(x, uo)
                ok = ok and K.res.info.status == Ko.res.info.status
                err2 = np.abs(K.output() - Ko.output()).max() / scale
                ok = ok and err <= 1e-6 and err2 <= 1e-6
                msg = 'err %.1e / %.1e' % (err, err2)
            else:
                msg = 'status ' + Ko.res.info.status
    except Exception as e:                                   # noqa: BLE001
        ok, msg = False, 'exception %r' % (e,)
    if not ok:
        bad += 1
    print(('ok   ' if ok else 'FAIL ') + tag + '  ' + msg, flush=True)
print('%d of %d failed' % (bad, count))
