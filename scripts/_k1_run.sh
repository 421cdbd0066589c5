cd $GRAFT_REPO_ROOT
timeout 120 python - <<'P' 2>&1 | grep -v amdgpu.ids | tail -12
import numpy as np, warnings
from pympc_amd.solver import BatchProblem
from pympc_amd import fixtures
warnings.simplefilter('ignore')
kw = fixtures.random_lti(3)
B = 8
bc = lambda v: np.broadcast_to(np.asarray(v, dtype=float), (B,) + np.shape(v))
res = {}
for be in ('sweeps', 'sweeps2'):
    p = BatchProblem(B, 12, 4, 30, backend=be, warm_start=1, eps_abs=1e-9, eps_rel=1e-9, max_iter=100000)
    p.setup(bc(kw['Ad']), bc(kw['Bd']), bc(kw['Qx']), bc(kw['QxN']), bc(kw['Qu']), bc(kw['QDu']), bc(kw['xmin']), bc(kw['xmax']), bc(kw['umin']), bc(kw['umax']), bc(kw['Dumin']), bc(kw['Dumax']), bc(kw['uref']), np.full((B, 1), kw['eps_feas']), bc(kw['x0']), bc(kw['uminus1']), bc(kw['xref']))
    p.solve_async(); p.synchronize()
    x, y, info = p.solution()
    print(be, p.kernel_name(False), p.occupancy(), [(i.status, i.iter) for i in info][:3], x[0, :3])
    res[be] = x.copy()
    xt, ut, st, it = p.mpc_run(5, w=0.01 * np.random.default_rng(0).standard_normal((5, B, 12)))
    print('  loop', st[-1][:4], it[-1][:4], ut[-1][0])
    res[be + 'u'] = ut
    p.close()
print('max |x diff|', np.abs(res['sweeps'] - res['sweeps2']).max(), 'max |u diff|', np.abs(res['sweepsu'] - res['sweeps2u']).max())
P
timeout 600 python scripts/shared_factor_rate.py --batch 1024 4096 --backend sweeps2 2>&1 | grep -v amdgpu.ids | cut -c1-250
