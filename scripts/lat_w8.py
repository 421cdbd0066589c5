"""ONE long-horizon controller of the reference's cart pole: MPCController.update() with the 512-thread kernel (AUTO) and with the 256-thread one
(mpcqp_settings.tuning = MPCQP_TUNE_NO_W8), the CPU oracle beside them:  python scripts/lat_w8.py [steps]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pympc_amd import MPCController, fixtures, _lib
from pympc_amd.solver import forced_settings
from oracle.osqp_oracle import OSQP
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for name, kw in (('notebook (4,1,150,75)', dict(fixtures.cart_pole(), Np=150, Nc=75)), ('kalman (4,1,200,200)', fixtures.cart_pole_kalman()), ('(4,1,100,100)', dict(fixtures.cart_pole(), Np=100)),
                 ('random (3,2,200,80)', dict(fixtures.random_lti(5, nx=3, nu=2, Np=200, xbox=4.0), Nc=80)), ('random (5,3,60,60)', fixtures.random_lti(6, nx=5, nu=3, Np=60, xbox=4.0)),
                 ('random (2,1,120,120)', fixtures.random_lti(7, nx=2, nu=1, Np=120, xbox=4.0))):
    res = {}
    for tag, tun in (('w8', 0), ('w4', _lib.TUNE_NO_W8), ('cpu', None)):
        with forced_settings(**({} if tun is None else dict(tuning=tun))), warnings.catch_warnings():
            warnings.simplefilter('ignore')
            K = MPCController(**kw)
            if tun is None:
                K.prob = OSQP()
            K.setup()
            x = np.array(kw['x0'], dtype=float); ts, us, its = [], [], []
            for _ in range(steps):
                u = K.output(); us.append(u.copy())
                x = kw['Ad'] @ x + kw['Bd'] @ u
                t = time.perf_counter(); K.update(x); ts.append(time.perf_counter() - t); its.append(K.res.info.iter)
            kn = K.prob.batch_problem.kernel_name(False) if tun is not None else 'oracle'
        res[tag] = (1e6 * np.median(ts), 1e6 * np.percentile(ts, 95), np.array(us), np.array(its), kn)
    print('%-22s w8 %6.1f us (p95 %6.1f) %s | w4 %6.1f us (p95 %6.1f) | cpu %6.1f us (p95 %6.1f) | same iteration counts %s / %s, |u8 - u4| %.1e |u8 - ucpu| %.1e' % (
        name, res['w8'][0], res['w8'][1], res['w8'][4], res['w4'][0], res['w4'][1], res['cpu'][0], res['cpu'][1],
        bool((res['w8'][3] == res['w4'][3]).all()), bool((res['w8'][3] == res['cpu'][3]).all()), np.abs(res['w8'][2] - res['w4'][2]).max(), np.abs(res['w8'][2] - res['cpu'][2]).max()), flush=True)
