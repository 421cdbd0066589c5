"""How full are the compute units during a device-loop launch of the headline batch?  python scripts/diag_makespan.py [batch] [steps] [--lib x.so] [--tuning 16 | 0]
From the per-instance entry / exit stamps of the last launch (mpcqp_get_launch_times): sum of the instances' residence times / #CU = the mean busy
time of a compute unit (one workgroup per CU at a time for the register-resident kernel), against the launch's length; the iteration counts per
instance give the time per iteration, early and late in the launch."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument('batch', type=int, nargs='?', default=1024); ap.add_argument('steps', type=int, nargs='?', default=20); ap.add_argument('--lib'); ap.add_argument('--tuning', type=int, default=16, help='mpcqp_settings.tuning (default 16 = TUNE_NO_QUEUE: the hardware dispatch this script was written to look at)')
a = ap.parse_args()
from pympc_amd import _lib
if a.lib: _lib.LIB_PATH = os.path.abspath(a.lib)
import numpy as np, torch
import bench
dev = torch.device('cuda', 0)
from pympc_amd.solver import forced_settings
with forced_settings(tuning=a.tuning):
    sh = bench.Shard(argparse.Namespace(eps=1e-3, chunk=None), bench.WORKLOADS['cfg3'][:4], a.batch, 0, 1, dev, 0, torch, None)
r = sh.measure('device_loop', a.steps, 5)
ll = r['last_launch']
t = ll['t'].astype(np.float64) * 1e-8
t0 = t[:, 0].min()
entry, leave = t[:, 0] - t0, t[:, 1] - t0
res = leave - entry
its = ll['its_step'].sum(axis=0)
wg, ncu, _ = sh.prob.occupancy()
print('batch %d, %d steps: launch %.3f ms (HIP events %.3f); sum of residence times / (%d CUs x %d) = %.3f ms -> %.1f %% of the slots\' time is busy'
      % (a.batch, a.steps, 1e3 * leave.max(), ll['ms'], ncu, wg, 1e3 * res.sum() / (ncu * wg), 100 * res.sum() / (ncu * wg) / leave.max()))
order = np.argsort(entry)
us_it = 1e6 * res / np.maximum(its, 1)
q = len(order) // 4 or 1
for name, idx in (('first quarter to start', order[:q]), ('second', order[q:2 * q]), ('third', order[2 * q:3 * q]), ('last', order[3 * q:])):
    print('  %-24s entry %.2f..%.2f ms  residence %.2f ms mean  %.0f iterations mean  %.3f us per iteration (incl. per-solve overheads)'
          % (name, 1e3 * entry[idx].min(), 1e3 * entry[idx].max(), 1e3 * res[idx].mean(), its[idx].mean(), us_it[idx].mean()))
print('  %.0f solves/s' % (a.batch * a.steps / r['elapsed']))
hw = sh.prob.launch_times(64)[:, -1]
if hw.any() and a.tuning & 16:          # (XCC / HW_ID of the workgroup that ran an instance's first steps: per-CU timelines make sense with one workgroup per instance only)
    xcc = (hw >> np.uint64(32)).astype(int) & 15; h = hw.astype(np.uint64) & np.uint64(0xFFFFFFFF); h = h.astype(np.int64)
    cu = (h >> 8) & 15; shh = (h >> 12) & 1; se = (h >> 13) & 7
    key = (xcc << 12) | (se << 8) | (shh << 4) | cu
    gaps, idle_end, per = [], [], []
    for k in np.unique(key):
        idx = np.where(key == k)[0]; idx = idx[np.argsort(entry[idx])]
        per.append(len(idx))
        for i0, i1 in zip(idx[:-1], idx[1:]): gaps.append(entry[i1] - leave[i0])
        idle_end.append(leave.max() - leave[idx[-1]])
    gaps = 1e6 * np.array(gaps); idle_end = 1e3 * np.array(idle_end)
    print('  %d distinct CUs; workgroups per CU min %d max %d; gap between two workgroups on a CU: median %.1f us, mean %.1f, p95 %.1f, max %.1f;  idle at the end: mean %.3f ms, max %.3f ms'
          % (len(per), min(per), max(per), np.median(gaps), gaps.mean(), np.percentile(gaps, 95), gaps.max(), idle_end.mean(), idle_end.max()))
    print('  workgroups per XCC:', np.bincount(xcc[np.argsort(entry)][:], minlength=8).tolist())
