"""Where a cfg-5 launch spends its time (development; optionally with a -DMPCQP_RUN_TIMING build through MPCQP_LIB):
  B=512 [STEPS=25] [LONE=1] [MPCQP_LIB=<lib>] python scripts/diag_cfg5_tail.py
  * the batch: kernel time of a STEPS-step device-loop launch, per-instance iteration totals (who are the stragglers),
  * LONE=1: ONE instance (the slowest of the batch) alone on the GPU: microseconds per plain ADMM iteration, per solve
    (begin + 25 iterations + check) and per closed-loop step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pympc_amd import _lib
if os.environ.get('MPCQP_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MPCQP_LIB'])
from pympc_amd import fixtures
from pympc_amd.solver import BatchProblem

NX, NU, NP, XBOX = (int(os.environ.get(k, d)) for k, d in (('NX', 20), ('NU', 8), ('NP', 100), ('XBOX', 1)))
B = int(os.environ.get('B', 512)); STEPS = int(os.environ.get('STEPS', 25)); EPS = float(os.environ.get('EPS', 1e-3))


def make(idx):
    kws = [fixtures.random_lti(i, nx=NX, nu=NU, Np=NP, xbox=float(XBOX)) for i in idx]
    st = lambda k: np.stack([np.asarray(kw[k], dtype=float) for kw in kws])
    n = len(idx)
    bp = BatchProblem(n, NX, NU, NP, eps_abs=EPS, eps_rel=EPS, warm_start=1)
    eye = lambda k, s: np.broadcast_to(s * np.eye(k), (n, k, k)).copy()
    ones = lambda k, s: np.full((n, k), s)
    bp.setup(st('Ad'), st('Bd'), eye(NX, 1.0), eye(NX, 1.0), eye(NU, 0.1), eye(NU, 0.1), ones(NX, -float(XBOX)), ones(NX, float(XBOX)),
             ones(NU, -1.0), ones(NU, 1.0), ones(NU, -0.5), ones(NU, 0.5), ones(NU, 0.0), np.full((n, 1), 1e6), st('x0'), ones(NU, 0.0), np.zeros((n, NX)))
    bp.solve_async(); bp.u0()
    return bp


def loop(bp, steps, seed):
    rng = np.random.default_rng(seed)
    w = 0.01 * rng.standard_normal((steps, bp.batch, NX))
    bp.stats(reset=True)
    bp.profile(enable=True, reset=True)
    out = bp.mpc_run(steps, w=w)
    bp.synchronize()
    ms, nl = bp.profile(enable=False)
    st = bp.stats(reset=True)
    return ms / max(1, nl), out[3], st


bp = make(range(B))
print('kernel', bp.kernel_name(True), 'B', B, file=sys.stderr)
loop(bp, STEPS, 1)
tot = None
for rep in range(3):
    ms, iters, st = loop(bp, STEPS, 2 + rep)
    per = iters.sum(axis=0)
    tot = per if tot is None else tot + per
    order = np.argsort(-per)
    print('launch %d: %.2f ms for %d steps (%.3f ms/step, %.0f solves/s)  iterations/solve %.2f  rounds/solve %.3f  slowest instances %s'
          % (rep, ms, STEPS, ms / STEPS, B * STEPS / ms * 1e3, st[0] / max(1, st[3]), st[1] / max(1, st[3]),
             [(int(i), int(per[i])) for i in order[:4]]))
print('per-instance iterations over the timed launches: min %d median %d max %d; instances above 1.5 x median: %d'
      % (tot.min(), np.median(tot), tot.max(), int((tot > 1.5 * np.median(tot)).sum())))
slow = int(np.argmax(tot))
del bp
if os.environ.get('LONE', '1') != '0':
    one = make([slow])
    loop(one, STEPS, 1)
    ms, iters, st = loop(one, STEPS, 2)
    print('instance %d ALONE: %.3f ms per closed-loop step, %.1f iterations/step, %.2f rounds/step' % (slow, ms / STEPS, st[0] / max(1, st[3]), st[1] / max(1, st[3])))
    one.profile(enable=True, reset=True)
    one.iterate(200)
    ms200, _ = one.profile(enable=False)
    one.profile(enable=True, reset=True)
    one.iterate(400)
    ms400, _ = one.profile(enable=False)
    us_it = 1e3 * (ms400 - ms200) / 200
    print('  plain iterations: %.1f us each (launch overhead + begin %.0f us)' % (us_it, 1e3 * ms200 - 200 * us_it))
    one.stats(reset=True)
    other = make([(slow + 1) % B])
    loop(other, STEPS, 1)
    ms, iters, st = loop(other, STEPS, 2)
    per_step = ms / STEPS
    its = st[0] / max(1, st[3]); rounds = st[1] / max(1, st[3])
    print('instance %d ALONE: %.3f ms per closed-loop step, %.1f iterations/step, %.2f rounds/step -> per step outside the iterations: %.0f us'
          % ((slow + 1) % B, per_step, its, rounds, 1e3 * per_step - its * us_it))
