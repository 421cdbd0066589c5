"""PCIe-inclusive rate: the cfg-3 batch driven with HOST (numpy) buffers through the C ABI --
x0 goes host->device, u* device->host every step, the plant runs in numpy on the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pympc_amd.solver import BatchProblem
B, nx, nu, Np = 1024, 12, 4, 30
d = bench.make_instances(bench.WORKLOADS['cfg3'][:4], 0, B)
prob = BatchProblem(B, nx, nu, Np, eps_abs=1e-3, eps_rel=1e-3)
eye = lambda k, s: np.broadcast_to(s * np.eye(k), (B, k, k))
ones = lambda k, s: np.full((B, k), s)
prob.setup(d['Ad'], d['Bd'], eye(nx, 1.0), eye(nx, 1.0), eye(nu, .1), eye(nu, .1), ones(nx, -10.), ones(nx, 10.), ones(nu, -1.), ones(nu, 1.),
           ones(nu, -.5), ones(nu, .5), ones(nu, 0.), np.full((B, 1), 1e6), d['x0'], ones(nu, 0.), np.zeros((B, nx)))
prob.solve_async(); u = prob.u0()
x = d['x0'].copy(); rng = np.random.default_rng(0)
def step():
    global x, u
    x = np.einsum('bij,bj->bi', d['Ad'], x) + np.einsum('bij,bj->bi', d['Bd'], u) + 0.01 * rng.standard_normal(x.shape)
    u = prob.mpc_step(x)                      # host pointers in and out: H2D 96 KB, D2H 32 KB per step
for _ in range(20): step()
t = time.perf_counter(); n = 100
t_host = 0.0
for _ in range(n):
    step()
el = time.perf_counter() - t
print('host-buffer (PCIe-inclusive) stepwise rate: %.0f QP-solves/s (%.3f ms per step of the batch, numpy plant included)' % (B * n / el, 1e3 * el / n))
