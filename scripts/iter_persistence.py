import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pympc_amd.solver import BatchProblem
B, nx, nu, Np = 1024, 12, 4, 30
d = bench.make_instances(bench.WORKLOADS['cfg3'][:4], 0, B)
prob = BatchProblem(B, nx, nu, Np, eps_abs=1e-3, eps_rel=1e-3)
eye = lambda k, s: np.broadcast_to(s * np.eye(k), (B, k, k))
ones = lambda k, s: np.full((B, k), s)
prob.setup(d['Ad'], d['Bd'], eye(nx, 1.0), eye(nx, 1.0), eye(nu, .1), eye(nu, .1), ones(nx, -10.), ones(nx, 10.), ones(nu, -1.), ones(nu, 1.),
           ones(nu, -.5), ones(nu, .5), ones(nu, 0.), np.full((B, 1), 1e6), d['x0'], ones(nu, 0.), np.zeros((B, nx)))
prob.solve_async(); prob.synchronize()
rng = np.random.default_rng(0)
w = 0.01 * rng.standard_normal((120, B, nx))
xt, ut, st, it = prob.mpc_run(120, w=w)
it = it[20:]                      # steady state
tot = it.sum(0)
print('per-instance total iterations over 100 steps: mean %.0f  min %d  max %d  p95 %.0f  p99 %.0f' % (tot.mean(), tot.min(), tot.max(), np.percentile(tot, 95), np.percentile(tot, 99)))
a, b = it[:50].sum(0), it[50:].sum(0)
print('correlation first/second half: %.3f' % np.corrcoef(a, b)[0, 1])
print('fraction of steps with 50 its: overall %.3f; per-instance quantiles' % (it == 50).mean(), np.percentile((it == 50).mean(0), [5, 25, 50, 75, 95]))
# CU load imbalance under the default placement: blocks b, b+256, b+512, b+768 share a CU
cu = tot.reshape(4, 256).sum(0)
print('per-CU work default placement: mean %.0f max %.0f (max/mean %.3f)' % (cu.mean(), cu.max(), cu.max() / cu.mean()))
order = np.argsort(-tot)
snake = np.concatenate([order[0:256], order[256:512][::-1], order[512:768], order[768:1024][::-1]])
cu2 = tot[snake].reshape(4, 256).sum(0)
print('per-CU work snake placement  : mean %.0f max %.0f (max/mean %.3f)' % (cu2.mean(), cu2.max(), cu2.max() / cu2.mean()))
