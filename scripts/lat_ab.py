"""The cyclic-reduction kernels side by side at small batches (BASELINE configs[3]: 128 instances per GPU when 1024 are split over 8):
  python scripts/lat_ab.py [--lib other.so] [--batches 128,256] [--backends bcr,bcrt,bcr8] [--steps 20 --warmup 5] [--timing]
per (backend, batch): QP-solves/s of the device loop (bench.py's Shard, same instances, same noise), the launch's critical path (slowest
instance's iterations x time per iteration), and the largest difference of the applied inputs from the first backend's.
--timing: the library is a -DMPCQP_RUN_TIMING build: print the phase clocks of ONE instance alone (mpcqp_get_stats writes them to stderr)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--lib'); ap.add_argument('--batches', default='128,256'); ap.add_argument('--backends', default='bcr,bcrt,bcr8')
ap.add_argument('--steps', type=int, default=20); ap.add_argument('--warmup', type=int, default=5); ap.add_argument('--timing', action='store_true')
ap.add_argument('--eps', type=float, default=1e-3)
a = ap.parse_args()
from pympc_amd import _lib
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
import numpy as np, torch
import bench
from pympc_amd.solver import forced_settings
dev = torch.device('cuda', 0)
dims = bench.WORKLOADS['cfg3'][:4]
args = argparse.Namespace(eps=a.eps, chunk=None)
for B in [int(b) for b in a.batches.split(',')]:
    ref = None
    for be in a.backends.split(','):
        with forced_settings(backend=be):
            sh = bench.Shard(args, dims, B, 0, 1, dev, 1000, torch, None)
        kn = sh.prob.kernel_name(True)
        r = sh.measure('device_loop', a.steps, a.warmup)
        u = sh.u.cpu().numpy()
        if ref is None:
            ref = u
        sp = r['launch_spread']
        print('batch %4d %-5s %-44s %9.0f solves/s  %.3f ms/step  kernel %.3f ms/launch  iters/solve %.1f  slowest instance %d its (median %d)  -> %.2f us per iteration of the slowest  |u - u_first| %.1e'
              % (B, be, kn, B * a.steps / r['elapsed'], 1e3 * r['elapsed'] / a.steps, r['run_ms'] / max(1, r['launches']), r['iters'] / max(1, r['solves']),
                 sp['iters_per_instance_max'], sp['iters_per_instance_median'], 1e3 * sp['ms_max'] / max(1.0, sp['iters_per_instance_max']) * (a.steps / sp['steps_per_launch'] if False else 1.0),
                 np.abs(u - ref).max()), flush=True)
        del sh
if a.timing:
    for be in a.backends.split(','):
        with forced_settings(backend=be):
            sh = bench.Shard(args, dims, 1, 0, 1, dev, 1000, torch, None)
        sh.measure('device_loop', 10, 5)
        sh.prob.stats(reset=True)
        r = sh.measure('device_loop', 20, 0)
        print('--- timing, one instance alone, backend %s: %d iterations in the timed launch' % (be, r['iters']), file=sys.stderr, flush=True)
        sh.prob.stats()
        del sh
