"""Phase clocks of ONE instance alone on its compute unit, -DMPCQP_RUN_TIMING builds:  python scripts/lat_phase.py <lib.so> [backend] [iters]
Prints cycles per ADMM iteration by phase (thread 0 of the workgroup: rhs | level 0 fwd | level 1 fwd | top | level 1 back | level 0 back | update)
and per round / per check outside the iterations.  The library writes its counters to stderr; this script reads them back through a pipe."""
import os, sys, re, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pympc_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
backend = sys.argv[2] if len(sys.argv) > 2 else 'bcr8'
import argparse, numpy as np, torch
import bench
from pympc_amd.solver import forced_settings
dev = torch.device('cuda', 0)
with forced_settings(backend=backend):
    sh = bench.Shard(argparse.Namespace(eps=1e-3, chunk=None), bench.WORKLOADS['cfg3'][:4], 1, 0, 1, dev, 1000, torch, None)
sh.measure('device_loop', 10, 5)
tmp = tempfile.TemporaryFile(mode='w+')
old = os.dup(2)
sh.prob.stats(reset=True)
r = sh.measure('device_loop', 20, 0)          # (its accounting reads -- and so prints and clears -- the counters once: captured)
os.dup2(tmp.fileno(), 2)
r2 = sh.measure('device_loop', 20, 0)
os.dup2(old, 2)
tmp.seek(0); txt = tmp.read()
its, rounds = r2['iters'], r2['checks']
m = [l for l in txt.splitlines() if 'iteration cycles' in l and 'total 1;' not in l][-1]
w = [l for l in txt.splitlines() if 'phase wall-clock' in l and 'admm 0' not in l][-1]
pct = [float(x) for x in re.findall(r'([0-9.]+)%', m)]
tot = float(re.search(r'total ([0-9.e+]+);', m).group(1))
outside = [float(x) for x in re.search(r't7 ([0-9.e+]+) t8 ([0-9.e+]+) t9 ([0-9.e+]+)', m).groups()]
chk = [float(x) for x in re.search(r'setup\+tail ([0-9.e+]+) rows ([0-9.e+]+) vars ([0-9.e+]+) reduce ([0-9.e+]+) decide ([0-9.e+]+)', m).groups()]
wall = [int(x) for x in re.search(r'begin (\d+) admm (\d+) check (\d+)', w).groups()]
names = ['rhs', 'L0fwd', 'L1fwd', 'top', 'L1back', 'L0back', 'update']
print('%s %s: %d iterations, %d rounds in 20 steps; %.0f cycles per iteration = ' % (os.path.basename(sys.argv[1]), backend, its, rounds, tot / its)
      + ' + '.join('%s %.0f' % (n, p / 100 * tot / its) for n, p in zip(names, pct)))
print('   whole admm function (thread 0): %.0f cycles per step; sum of its slots: %.0f; before the first tick %.0f' % (chk[1] / 20.0, (tot + sum(outside)) / 20.0, chk[2] / 20.0))
print('   per round: load %.0f  owner regs %.0f  write-back %.0f cycles;  per check: setup+tail %.0f vars %.0f reduce %.0f decide %.0f cycles;  wall (10 ns ticks) per step: begin %.0f admm %.0f check %.0f'
      % tuple([o / rounds for o in outside] + [chk[0] / rounds, chk[2] / rounds, chk[3] / rounds, chk[4] / rounds] + [x / 20.0 for x in wall]))
