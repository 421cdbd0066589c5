#!/usr/bin/env python3
"""bench.py -- QP-solves/s of the MPC hot path on MI355X (BASELINE.json metric) and max |u* - u*_ref|.

One "step" = one closed-loop MPC step of the whole batch: plant update x+ = Ad x + Bd u* + w,
QP refresh (q, l, u from the new x0 and u_{-1}) and one warm-started ADMM solve of every instance, u* fetched.
Workload (BASELINE.json configs[2]): 1024 seed-pinned random stable LTI
systems nx=12, nu=4, Np=30 per GPU, reference-default tolerances (eps_abs=eps_rel=1e-3,
pyMPC/mpc.py:80), synthetic data, FP64, all inputs resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 100 --warmup 20
    python bench.py --gpus N ...                   # WORLD_SIZE unset: re-executes itself under torch.distributed.run with N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      # what the driver runs
    python bench.py --workload cfg5                # BASELINE configs[4]: 512 x (20,8,100), Delta-u + slack rows active
    python bench.py --workload cfg2                # BASELINE configs[1]: ONE cart-pole controller (4,1,20), latency per update()
    python bench.py --workload notebook            # examples/example_inverted_pendulum_kalman.ipynb shape (4,1,150,75), one controller
    ... bench.py --gpus N --total-batch 1024       # BASELINE configs[3] read as strong scaling: the SAME 1024 instances over N GPUs

Two ways through the same library, both measured, `--path` chooses which one is `value` (the other is `other_path`):
  device_loop (default): the K steps run inside mpcqp_mpc_loop launches (output -> plant -> update -> solve per
                         instance on the device), at most 50 steps per launch (--chunk);
  stepwise             : the reference's call pattern, update()/solve()/output() per step from the host.
Both give bit-identical trajectories (tests/test_gpu_parity.py::test_device_loop_*).

Rank 0 prints ONE JSON line.  With N > 1 the instances are sharded over ranks (weak scaling: `--batch` per GPU; strong
scaling with --total-batch); RCCL is used only to scatter the problem data from rank 0 and to all-gather u*; every rank
reports (rank, device identity) and the line carries `ranks_seen` / `devices_seen` (the run fails if either is < N).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md chip-level parameters (achievable: ~6.3e12)
HBM_ACHIEVABLE = 6.29e12   # B/s, the same guide's measured streaming rate (float4 copy, 79 % of the spec figure)
# Whole-chip READ-ONLY streams shaped like the factor stream (1024 workgroups x 4 waves, 32 bytes per lane; scripts/diag/hbm_stream.hip,
# profiles/r5_hbm_stream.txt): 5.9-6.0 TB/s from HBM (1 and 4 GB buffers), 6.9 TB/s from the Infinity Cache (128 MB); a copy of the same shape 5.0 / 6.2
HBM_READ_STREAM = 5.95e12
MALL_READ_STREAM = 6.90e12
MFMA_F64_PEAK = 78.6e12    # flop/s, FP64 matrix peak of MI355X (AMD datasheet; scripts/diag/mfma_rate.hip: 17 cycles per v_mfma_f64_4x4x4_4b = 74e12 measured)
INFINITY_CACHE = 256 << 20  # bytes, MI355X_MICROARCH.md (memory-side cache in front of HBM)
U_ERR_SAMPLE = 32          # instances whose u* is compared with the tight-tolerance CPU reference
WORKLOADS = {              # name -> (nx, nu, Np, state box, default instances per GPU)
    'cfg3': (12, 4, 30, 10.0, 1024),
    'cfg5': (20, 8, 100, 1.0, 512),
}


# ----------------------------------------------------------------------------------------------------------------------
# launching: one process per GPU
# ----------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch_if_needed(args):
    """`python bench.py --gpus N` with N > 1 outside torch.distributed.run: start the N ranks ourselves (same interpreter,
    same flags, rendezvous on 127.0.0.1) and hand back their exit code.  Inside a launcher (WORLD_SIZE set) this is a no-op."""
    if args.gpus <= 1 or 'WORLD_SIZE' in os.environ:
        return
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: RCCL needs it on this host driver
    sys.exit(subprocess.call(cmd, env=env))


def comm_on(world):
    """Does this run talk through the process group?  Always with more than one rank; with ONE rank only when MPCQP_BENCH_FORCE_PG=1 asks for
    it -- the way to execute the RCCL branch (process group on the device, barrier, max-reduction of the time, all-gather of u*) on a box that
    has a single GPU (tests/test_gpu_rccl_single_rank.py): same code path, a communicator of one."""
    return world > 1 or os.environ.get('MPCQP_BENCH_FORCE_PG') == '1'


def device_identity(torch, dev):
    """Something that distinguishes physical GPUs: the device UUID if this torch exposes it, else the PCI address."""
    if dev.type != 'cuda':
        return 'cpu:%s:%d' % (socket.gethostname(), os.getpid())
    p = torch.cuda.get_device_properties(dev)
    uuid = getattr(p, 'uuid', None)
    if uuid is not None:
        return 'uuid:%s' % uuid
    pci = tuple(getattr(p, k, None) for k in ('pci_domain_id', 'pci_bus_id', 'pci_device_id'))
    if any(v is not None for v in pci):
        return 'pci:%s:%s:%s' % pci
    return 'index:%d:%s' % (dev.index, p.name)


def flush_c_stdio():
    """RCCL prints a banner (version, host, library path) through C stdio when a communicator is made; on a pipe that buffer is written when the process exits --
    AFTER the JSON line, which must be the LAST line on stdout.  Push it out now (every rank: under a launcher the ranks share one stdout)."""
    try:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:      # noqa: BLE001  (no libc handle: nothing to flush through it)
        pass


def init_distributed(args, torch, dist):
    """Process group + the proof that N ranks on N distinct devices are in it.  Returns (rank, world, local_rank, dev, seen)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('MPCQP_BENCH_BACKEND', 'nccl')     # ('gloo': the CPU test of this plumbing, with --dry-run)
    if args.dry_run and backend == 'gloo':
        dev = torch.device('cpu')
    else:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an AMD GPU; pympc_amd has no CPU fallback')
        if world > 1 and torch.cuda.device_count() < world:
            raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible' % (world, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    if args.gpus != world and rank == 0:
        print('bench.py: --gpus %d but WORLD_SIZE=%d; running with %d rank(s)' % (args.gpus, world, world), file=sys.stderr)
    seen = [(rank, device_identity(torch, dev))]
    if comm_on(world):
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(free_port()))      # (a forced single rank outside a launcher)
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=dev)
        else:
            dist.init_process_group(backend=backend)
        got = [None] * world
        dist.all_gather_object(got, seen[0])
        seen = sorted(tuple(g) for g in got)
        dist.barrier()
        flush_c_stdio()                                          # (the communicator exists now: its banner goes out here, not behind the JSON line)
    ranks_seen = len({r for r, _ in seen})
    devices_seen = len({d for _, d in seen})
    if ranks_seen < world or devices_seen < world:
        raise SystemExit('bench.py: %d rank(s) on %d distinct device(s) in the process group, expected %d of each: %r'
                         % (ranks_seen, devices_seen, world, seen))
    return rank, world, local_rank, dev, dict(ranks_seen=ranks_seen, devices_seen=devices_seen,
                                              backend=backend if comm_on(world) else None, devices=[d for _, d in seen])


# ----------------------------------------------------------------------------------------------------------------------
# workload
# ----------------------------------------------------------------------------------------------------------------------
def make_instances(dims, first, count):
    from pympc_amd import fixtures
    nx, nu, Np, xbox = dims
    kws = [fixtures.random_lti(first + i, nx=nx, nu=nu, Np=Np, xbox=xbox) for i in range(count)]
    return {k: np.stack([np.asarray(kw[k], dtype=float) for kw in kws]) for k in ('Ad', 'Bd', 'x0')}


def algorithmic_bytes_8d(dims, n, m, nnzL, iters, checks, solves):
    """a generic sparse-LDL' ADMM, FP64 values only.  per iteration 8(2 nnzL_strict + 6n + 10m); per
    residual evaluation 8(nnz(triu P) + 2 nnz(A)); per solve 8(4n + 6m).  Kept beside the design figure: it charges the
    iterate and metric vectors to HBM every iteration, which this implementation keeps in LDS/registers."""
    NX, NU, NP, _ = dims
    nnz_triuP = (NP + 1) * NX * 2 + NP * NU + (NP - 1) * NU     # diagonal weights: diag + upper QDu coupling
    nnzA = (NP + 1) * NX + NP * NX * NX + NP * NX * NU + 2 * (NP + 1) * NX + NP * NU + NU + 2 * NP * NU - 1
    b_it = 8 * (2 * (nnzL - n) + 6 * n + 10 * m)
    return iters * b_it + checks * 8 * (nnz_triuP + 2 * nnzA) + solves * 8 * (4 * n + 6 * m), b_it


def pmc_entry(workload, path, kernel):
    """The committed rocprofv3 PMC summary of this same command (profiles/pmc_hbm_traffic.json: FETCH_SIZE and WRITE_SIZE
    collected in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, divided by the ADMM
    iterations of the profiled launches).  Counters cannot be read from inside the process; the entry records the batch it
    was profiled with and is only used for a run of the same shape."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_hbm_traffic.json')) as f:
            return json.load(f)[workload][path][kernel]
    except Exception:
        return None


def osqp_available():
    try:
        import osqp  # noqa: F401
        return True
    except Exception:
        return False


def real_osqp_pin():
    """If the REAL solver the reference calls (PyPI `osqp`, pyMPC/mpc.py:4) is importable on this box, run the comparisons of
    tests/test_real_osqp.py inline (oracle vs osqp at the default tolerance: status / iterations / iterate; osqp vs the
    certified optimum goldens at 1e-10) and report -- the first box that has the wheel pins the oracle without code changes."""
    if not osqp_available():
        return {'osqp_available': False}
    out = {'osqp_available': True}
    try:
        import osqp
        import scipy.sparse as sp
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from util import golden_names, load_golden, golden_csc
        from oracle.osqp_oracle import OSQP as Oracle
        out['osqp_version'] = getattr(osqp, '__version__', '?')
        res = {}
        for name in golden_names():
            g = load_golden(name)
            P, A = golden_csc(g, 'P'), golden_csc(g, 'A')
            pr = osqp.OSQP()
            pr.setup(sp.triu(P, format='csc'), np.array(g['q']), A.tocsc(), np.array(g['l']), np.array(g['u']), verbose=False,
                     eps_abs=1e-3, eps_rel=1e-3, adaptive_rho_interval=100)
            r = pr.solve()
            o = Oracle()
            o.setup(P, g['q'], A, g['l'], g['u'], eps_abs=1e-3, eps_rel=1e-3, adaptive_rho_interval=100)
            ro = o.solve()
            res[name] = dict(status_equal=ro.info.status == r.info.status, iter_oracle=int(ro.info.iter), iter_osqp=int(r.info.iter),
                             x_err=float(np.abs(ro.x - r.x).max()) if r.x is not None and r.x[0] is not None else None)
        out['oracle_vs_osqp_eps1e-3'] = res
        out['all_equal'] = all(v['status_equal'] and v['iter_oracle'] == v['iter_osqp'] for v in res.values())
    except Exception as e:
        out['error'] = repr(e)
    return out


def cpu_legs(dims, eps, samples, want_baseline, seconds_budget=12.0, inst=400, steps=100):
    """Everything that needs the CPU oracle (test infrastructure; the only place bench.py touches oracle/):
    (1) cpu_baseline -- the reference-style CPU path on this box's host cores: the C port of the OSQP algorithm
        (oracle/osqp_ref.c, rebuilt here with -O3 -march=native) driven by its C closed-loop driver like the reference
        drives OSQP: 1 thread, sequential over instances, warm-started receding horizon on the same workload recipe
        (`value`, what pyMPC does today), and the same on every usable core at once;
    (2) u*_ref of the sampled QPs at tolerance 1e-10, for 'max |u* - u*_ref|'."""
    from oracle import cpu_bench
    NX, NU, NP, XBOX = dims
    out, refs = None, []
    if want_baseline:
        cpu_bench.build_native()
        n_solve, t_solve, iters, done, flags = cpu_bench.in_subprocess(cpu_bench.run_instances, (0, inst, steps, eps, NX, NU, NP, XBOX, seconds_budget))
        out = dict(value=n_solve / t_solve, unit='QP-solves/s', cores=1, kind='port',
                   sample='%d instances x %d warm-started steps of the same workload (oracle/osqp_ref.c, %s, C closed-loop driver: update+solve only, '
                          'mean %.1f ADMM iterations/solve, %.1f s of CPU work)' % (done, steps, flags, iters / max(1, n_solve), t_solve))
        try:
            a = cpu_bench.all_cores(steps, eps, NX, NU, NP, XBOX, seconds_budget)
            out['all_cores'] = dict(value=a['value'], unit='QP-solves/s', cores=a['cores'], kind='port',
                                    sample='%d instances x %d steps over %d worker processes, %.1f CPU-seconds, mean %.1f iterations/solve'
                                           % (a['instances'], steps, a['cores'], a['cpu_seconds'], a['mean_iters']))
        except Exception as e:          # the single-core figure stands on its own
            out['all_cores'] = {'error': repr(e)}
    for s in samples:
        refs.append(cpu_bench.in_subprocess(cpu_bench.reference_inputs, (s['idx'], s['x0'], s['um1'], NX, NU, NP, XBOX)))
    return out, refs


class Shard:
    """One rank's share of the batch: B instances set up on this GPU, and the two measured ways through the library."""

    def __init__(self, args, dims, B, rank, world, dev, first_instance, torch, dist, total=None):
        from pympc_amd.solver import BatchProblem
        from pympc_amd import sharding
        # total: instances over all ranks (default world * B); with a total that does not divide, the last rank(s) are short (sharding.shard_range)
        total = B * world if total is None else total                # (B may be None when total is given)
        lo, hi = sharding.shard_range(total, rank, world)
        B = hi - lo
        if B < 1:
            raise SystemExit('bench.py: rank %d of %d has no instance of the %d to work on (more GPUs than blocks of %d)' % (rank, world, total, sharding.shard_rows(total, world)))
        self.total, self.first_local = total, lo
        self.args, self.dims, self.B, self.rank, self.world, self.dev = args, dims, B, rank, world, dev
        self.torch, self.dist, self.sharding = torch, dist, sharding
        NX, NU, NP, XBOX = dims
        f64 = torch.float64
        # ---- problem data: generated on rank 0, scattered over RCCL (north_star: scatter problem data)
        full = None
        if rank == 0:
            full = {k: torch.from_numpy(v).to(dev) for k, v in make_instances(dims, first_instance, total).items()}
        self.comm = comm_on(world)
        if self.comm:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.shared = bool(getattr(args, 'shared_model', False))
        if self.shared:
            # ONE model for every instance of every rank (last paragraph; test_scripts/example_mpc_function.py:105-111): the first instance's model and
            # state are broadcast once, the states alone are scattered; every rank sets its shard up as copies of that ONE controller (its instances then
            # share one KKT factor, mpcqp_share_factor) and update() scatters the states
            mdl = sharding.broadcast_model({k: full[k][0] for k in ('Ad', 'Bd', 'x0')} if rank == 0 else None, {'Ad': (NX, NX), 'Bd': (NX, NU), 'x0': (NX,)}, dev)
            loc = sharding.scatter_instances({'x0': full['x0']} if rank == 0 else None, {'x0': (NX,)}, None, dev, total=total)
            loc['Ad'], loc['Bd'] = mdl['Ad'].expand(B, NX, NX).contiguous(), mdl['Bd'].expand(B, NX, NU).contiguous()
            x_setup = mdl['x0'].expand(B, NX).contiguous()
        else:
            loc = sharding.scatter_instances(full, {'Ad': (NX, NX), 'Bd': (NX, NU), 'x0': (NX,)}, None, dev, total=total)      # one packed collective
        torch.cuda.synchronize()
        self.scatter_ms = 1e3 * (time.perf_counter() - t0)
        self.gather_ms, self.gathers = 0.0, 0
        self.Ad, self.Bd, self.x = loc['Ad'], loc['Bd'], loc['x0'].clone()
        if not self.shared:
            x_setup = self.x
        stream = torch.cuda.current_stream(dev)
        self.prob = prob = BatchProblem(B, NX, NU, NP, device=dev.index, stream=stream.cuda_stream,
                                        eps_abs=args.eps, eps_rel=args.eps, warm_start=1)
        eye = lambda k, s: (s * torch.eye(k, dtype=f64, device=dev)).expand(B, k, k).contiguous()
        ones = lambda k, s: torch.full((B, k), s, dtype=f64, device=dev)
        # SURVEY 8(d) measurement (i): cold setup() + first solve (pyMPC/mpc.py:254-269), timed with events on the stream
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        setup_args = (self.Ad, self.Bd, eye(NX, 1.0), eye(NX, 1.0), eye(NU, 0.1), eye(NU, 0.1),
                      ones(NX, -XBOX), ones(NX, XBOX), ones(NU, -1.0), ones(NU, 1.0), ones(NU, -0.5), ones(NU, 0.5),
                      ones(NU, 0.0), torch.full((B, 1), 1e6, dtype=f64, device=dev),
                      x_setup, ones(NU, 0.0), torch.zeros((B, NX), dtype=f64, device=dev))      # (built first: torch's fill kernels load lazily, ~100 ms the first time)
        torch.cuda.synchronize()
        ev[0].record()
        prob.setup(*setup_args)
        ev[1].record()
        self.u = torch.empty((B, NU), dtype=f64, device=dev)
        prob.solve_async()                        # cold solve (setup(solve=True))
        prob.u0(out=self.u)                       # (a solve may finish in a second launch that the first call reading its results issues)
        ev[2].record()
        torch.cuda.synchronize()
        st = prob.stats(reset=True)
        # ... and the same two calls once more: the first setup() of a process also loads the library's code object and its kernels (one-time, ~ 3 ms)
        ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev2[0].record()
        prob.setup(*setup_args)
        ev2[1].record()
        prob.solve_async()
        prob.u0(out=self.u)
        ev2[2].record()
        torch.cuda.synchronize()
        st2 = prob.stats(reset=True)
        self.instances_sharing = None
        if self.shared:
            self.instances_sharing = prob.share_factor()      # (setup has built the map already; this reports it)
            prob.update(self.x, ones(NU, 0.0))                # the scattered states
            prob.solve_async()
            prob.u0(out=self.u)
            torch.cuda.synchronize()
        self.cold = dict(setup_ms=ev[0].elapsed_time(ev[1]), first_solve_ms=ev[1].elapsed_time(ev[2]),
                         setup_ms_repeat=ev2[0].elapsed_time(ev2[1]), first_solve_ms_repeat=ev2[1].elapsed_time(ev2[2]), repeat_same_iterations=bool(st2[0] == st[0]),
                         iters_per_instance=st[0] / B, refactorizations_per_instance=st[2] / B, instances=B,
                         note='mpc.py:254-269: setup() = upload + QP build + 10 Ruiz passes + first factorization of every instance '
                              '(host upload included), first_solve = cold-started ADMM solve incl. its rho-update refactorizations')
        # whole-process work per kernel (for the profile scripts: counter totals / these = bytes per iteration): 'solve' =
        # k_mpc_run<..,false> (mpcqp_solve), 'loop' = k_mpc_run<..,true> (mpcqp_mpc_loop)
        self.totals = {'solve': [0, 0, 0], 'loop': [0, 0, 0]}
        self.account('solve', st)
        self.account('solve', st2)
        # process noise w_k ~ N(0, 0.01^2): instance i draws from ITS OWN seed-pinned stream (fixtures.random_lti_noise_rng, SURVEY 8d:
        # "w from the same rng" recipe) -- the same realisation the CPU baseline's closed loop uses (oracle/cpu_bench.py), whatever
        # the batch size, the rank count or the path; rows are consumed in step order across all the measurements of this shard
        from pympc_amd import fixtures
        self.noise_rngs = [fixtures.random_lti_noise_rng(first_instance + lo + j) for j in range(B)]
        self.per = sharding.shard_rows(total, world)
        self.u_all = torch.empty((world * self.per, NU), dtype=f64, device=dev) if self.comm else None

    def account(self, kind, st=None):
        st = self.prob.stats(reset=True) if st is None else st
        for i, v in enumerate((st[0], st[1], st[3])):
            self.totals[kind][i] += v
        return st

    def noise(self, count):
        """[count, B, nx] device tensor: the next `count` rows of every instance's noise stream."""
        nx = self.dims[0]
        w = np.stack([0.01 * r.standard_normal((count, nx)) for r in self.noise_rngs], axis=1)
        return self.torch.from_numpy(w).to(self.dev)

    def plant(self, xc, uc, w):
        torch = self.torch
        return torch.baddbmm(w.unsqueeze(2), self.Ad, xc.unsqueeze(2)).add_(torch.bmm(self.Bd, uc.unsqueeze(2))).squeeze(2)

    def timed(self, kind, run_warm, run_timed):
        """W untimed steps, then the timed K steps between barrier + synchronize; returns max-over-ranks seconds
        and the device-side accounting of the timed region."""
        torch, dist, prob, world = self.torch, self.dist, self.prob, self.world
        run_warm()
        self.account(kind)
        prob.profile(enable=True, reset=True)
        if self.comm:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_timed()
        if self.comm:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if self.comm:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed_local, elapsed = elapsed, float(t.item())
        else:
            elapsed_local = elapsed
        iters, checks, refacts, solves = self.account(kind)
        run_ms, launches = prob.profile(enable=False)
        return dict(elapsed=elapsed, elapsed_local=elapsed_local, iters=iters, checks=checks, refacts=refacts, solves=solves, run_ms=run_ms, launches=launches)

    def timed_gather(self, fn):
        """An all-gather of u* with its wall time on this rank (host clock around a synchronised call: the exchange is 32 B per instance and
        step, its cost is latency) -- reported per rank beside the scatter."""
        torch = self.torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        self.gather_ms += 1e3 * (time.perf_counter() - t0); self.gathers += 1

    def measure_stepwise(self, steps, warmup):
        """The reference's call pattern: per step the host calls update(), solve(), output() (one kernel launch per
        solve); plant and disturbance are torch ops on the same stream; with N > 1 u* is all-gathered every step."""
        w_all = iter(self.noise(warmup + steps))       # synthetic input, generated before the timed region

        def step():
            self.x = self.plant(self.x, self.u, next(w_all))
            self.prob.update(self.x, self.u)
            self.prob.solve_async()
            self.prob.u0(out=self.u)
            if self.comm:
                self.timed_gather(lambda: self.sharding.gather_inputs(self.u, out=self.u_all, total=self.total))

        return self.timed('solve', lambda: [step() for _ in range(warmup)], lambda: [step() for _ in range(steps)])

    def measure_device_loop(self, steps, warmup):
        """The same closed loop inside mpcqp_mpc_loop: launches of `chunk` steps each, so that every
        launch (warm-up and timed) does the same work; the disturbance sequence is synthetic input generated before
        the timed region; with N > 1 the applied inputs of a chunk are all-gathered after it."""
        torch, B, dev, world = self.torch, self.B, self.dev, self.world
        NX, NU = self.dims[0], self.dims[1]
        f64 = torch.float64
        # steps per launch: a launch ends when its slowest instance has finished its steps (no instance can run ahead of its
        # own closed loop), so short launches pay the spread of the per-instance iteration counts more often -- at cfg-3,
        # 5-step launches cost 8 % against 20-step ones, and at cfg-5 50-step launches gain 2-3 % on 25-step ones (per-instance iteration totals
        # average out over a longer launch).  Default: the whole timed region in equal launches of at most 50 steps.
        if self.args.chunk:
            chunk = self.args.chunk
        else:
            chunk = steps
            while chunk > 50:
                chunk = next((chunk // d for d in (2, 3, 5, 7) if chunk % d == 0), 50)
        w_all = self.noise(warmup + steps)
        outs = (torch.empty((chunk + 1, B, NX), dtype=f64, device=dev), torch.empty((chunk, B, NU), dtype=f64, device=dev),
                torch.empty((chunk, B), dtype=torch.int32, device=dev), torch.empty((chunk, B), dtype=torch.int32, device=dev))
        u_hist = torch.empty((world * chunk, self.per, NU), dtype=f64, device=dev) if self.comm else None

        # per launch: an event pair on the launch stream (the handle's stream IS torch's current stream) and the instances' iteration
        # counts, summed on the device (one tiny reduction per launch, inside the timed region: extra work, nothing skipped)
        marks, it_sum = [], torch.zeros((B,), dtype=torch.int64, device=dev)
        it_last = torch.zeros((B,), dtype=torch.int64, device=dev)
        it_sum.add_(outs[3].sum(dim=0)); it_sum.zero_()            # (torch loads a kernel the first time it is used, ~100 ms: not inside the timed region)
        it_last.copy_(outs[3].sum(dim=0)); it_last.zero_()

        def run(first, count, record):
            o = None
            for c in range(first, first + count, chunk):
                k = min(chunk, first + count - c)
                o = outs if k == chunk else tuple(t[:k + (1 if i == 0 else 0)] for i, t in enumerate(outs))
                if record:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                self.prob.mpc_run(k, w=w_all[c:c + k], out=o)
                if record:
                    e1.record(); marks.append((e0, e1, k))
                    it_last.copy_(o[3].sum(dim=0))
                    it_sum.add_(it_last)
                if self.comm and k == chunk:
                    self.timed_gather(lambda: self.sharding.gather_trajectory(outs[1], out=u_hist, total=self.total))
            return o

        last = {}
        r = self.timed('loop', lambda: last.update(o=run(0, warmup, False)) if warmup else None, lambda: last.update(o=run(warmup, steps, True)))
        self.x = last['o'][0][-1].clone()
        self.u.copy_(last['o'][1][-1])
        r['chunk'] = chunk
        ms = [e0.elapsed_time(e1) for e0, e1, _ in marks]
        # the LAST timed launch, instance by instance: when its workgroup entered and left (the kernel stamps the chip's 100 MHz clock) and how many
        # iterations it ran -- what roofline() turns into the streaming rate BEFORE the launch's tail (frac_excluding_tail)
        kl = marks[-1][2]
        r['last_launch'] = {'t': self.prob.launch_times(min(kl, 64)), 'its_step': last['o'][3][:kl].cpu().numpy().astype(float), 'steps': kl, 'ms': ms[-1]}
        its = it_sum.cpu().numpy().astype(float)
        med = float(np.median(its))
        # a launch ends with its slowest instance (no instance can run ahead of its own closed loop): how uneven the work was
        r['launch_spread'] = {'launches': len(ms), 'steps_per_launch': chunk, 'ms_min': min(ms), 'ms_max': max(ms), 'ms_mean': float(np.mean(ms)),
                              'iters_per_instance_median': med, 'iters_per_instance_max': float(its.max()),
                              'instances_above_1p5x_median': int((its > 1.5 * med).sum()),
                              'critical_path_ratio': float(its.max() / max(1.0, med))}
        return r

    def measure(self, path, steps, warmup):
        return {'stepwise': self.measure_stepwise, 'device_loop': self.measure_device_loop}[path](steps, warmup)

    def sample_point(self, nsample=U_ERR_SAMPLE):
        """One more (untimed) closed-loop step through the stepwise API: returns the sampled QPs (x0, u_{-1}) and the
        u* the device produced for them at the current tolerance.  Rank 0's instances only (global index = local)."""
        torch, B = self.torch, self.B
        self.x = self.plant(self.x, self.u, self.noise(1)[0])
        um1 = self.u.clone()
        self.prob.update(self.x, um1)
        self.prob.solve_async()
        self.prob.u0(out=self.u)
        torch.cuda.synchronize()
        self.account('solve')
        idx = np.unique(np.linspace(0, B - 1, min(nsample, B)).astype(int))
        infos = self.prob.infos()
        return dict(idx=idx, x0=self.x[idx].cpu().numpy(), um1=um1[idx].cpu().numpy(), u=self.u[idx].cpu().numpy(),
                    solved=np.array([infos[int(i)].status == 1 for i in idx]))

    def refactor_ms(self):
        torch, prob = self.torch, self.prob
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        prob.refactor(); torch.cuda.synchronize()
        ev0.record()
        for _ in range(3):
            prob.refactor()
        ev1.record(); torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / 3

    def working_set_bytes(self, active_instances=None):
        """Bytes the timed loop touches per GPU: every instance's KKT factor, iterate, metric, model and step data.
        active_instances: only that many instances are in flight at a time (the device loop: one per resident workgroup slot -- an instance runs
        its whole K-step closed loop before the slot takes the next one, so the ACTIVE set is what the caches see)."""
        p = self.prob
        per = 8 * (p.factor_doubles + 3 * p.n + 5 * p.m + 2 * (p.n + p.m)) + 8 * 1024
        return int(per * (self.B if active_instances is None else min(self.B, active_instances)))

    @staticmethod
    def tail_split(res, per_iter, per_round, per_solve, check_every=25):
        """A launch ends with its slowest instance: the compute units are full only for part of it.  From the last timed launch: every instance
        stamps the chip's 100 MHz clock when its workgroup enters, after each closed-loop step, and when it leaves (mpcqp_get_launch_times); the
        design bytes of a step of instance i are its_ij per_iter + (its_ij / 25) per_round + per_solve, moved between the end of its previous step
        and the end of this one.  t10 = the moment 10 % of the instances have left (90 % of the slots still busy); bytes_before = the bytes of all
        steps completed by then plus the running steps' share in proportion of time.  Returns bytes_before / t10 in B/s and the landmark times."""
        ll = res.get('last_launch')
        if not ll or not ll['t'].any():
            return None
        t = ll['t'].astype(np.float64) * 1e-8              # 100 MHz ticks -> seconds; [B, 2 + k]
        t0 = t[:, 0].min()
        entry, leave = t[:, 0] - t0, t[:, 1] - t0
        k = t.shape[1] - 2
        ends = t[:, 2:] - t0                                # [B, k]
        begins = np.concatenate([entry[:, None], ends[:, :-1]], axis=1)
        its = ll['its_step'].T[:, :k]                       # [B, k]
        by = its * per_iter + its / check_every * per_round + per_solve
        q = lambda f: float(np.quantile(leave, f))
        t10 = q(0.10)
        share = np.clip((t10 - begins) / np.maximum(ends - begins, 1e-12), 0.0, 1.0)
        before = float((by * share).sum())
        return {'rate_before_tail': before / t10, 'bytes_before_tail': before, 'bytes_launch': float(by.sum()),
                'ms_10pct_left': 1e3 * t10, 'ms_50pct_left': 1e3 * q(0.5), 'ms_90pct_left': 1e3 * q(0.9), 'ms_all_left': 1e3 * float(leave.max()),
                'ms_launch_hip_events': ll['ms'], 'steps': ll['steps'],
                'definition': 'last timed launch: per-instance stamps of the chip\'s 100 MHz clock at workgroup entry, after every closed-loop step and at exit '
                              '(mpcqp_get_launch_times) with the per-step iteration counts give the design bytes moved by any moment to within one step of one instance; '
                              'rate_before_tail = bytes moved until 10 % of the instances have left / that time'}

    def roofline(self, res, path, workload_key):
        """HBM roofline of the one kernel of the path, k_mpc_run (QP refresh, ADMM iterations, residual checks).  HIP events
        bracket every launch on its stream (mpcqp_profile); `achieved` divides the bytes this implementation streams BY
        DESIGN (mpcqp_get_stream_bytes x the device-side iteration / round / solve counters of the timed region) by that time."""
        prob = self.prob
        iters, checks, solves = res['iters'], res['checks'], res['solves']
        admm_ms, launches = res['run_ms'], max(1, res['launches'])
        per_iter, per_round, per_solve = prob.stream_bytes()
        shared_frac = (self.instances_sharing or 0) / self.B if getattr(self, 'shared', False) else 0.0
        if shared_frac:
            # copies of ONE controller on a shared factor (mpcqp_share_factor): the factor stream of the sharing instances is ONE 8 * factor_doubles block that stays
            # in L2 -- not memory traffic, and no committed counter profile is of this command (profiles/r6_shared_factor_fetch.txt: 312 -> 9.4 GB per launch)
            per_iter = max(0, per_iter - int(shared_frac * 8 * prob.factor_doubles))
            workload_key = None
        design_bytes = iters * per_iter + checks * per_round + solves * per_solve
        achieved = design_bytes / (admm_ms * 1e-3)
        alg8d, b_it8d = algorithmic_bytes_8d(self.dims, prob.n, prob.m, prob.nnzL, iters, checks, solves)
        kname = prob.kernel_name(loop=path == 'device_loop')
        lds_state = ',true,' in kname.split('<')[1][:9]            # second template argument: iterate resident in LDS
        pmc = pmc_entry(workload_key, path, kname) if workload_key else None      # (keys: cfg3, cfg5, cfg3_b4096 -- one per profiled command)
        pmc_batch = (pmc or {}).get('batch')
        pmc_ok = pmc is not None and (pmc_batch is None or pmc_batch == self.B)
        pmc_b = pmc['hbm_bytes_per_iter_per_qp'] if pmc_ok else None
        traffic = pmc_b * iters / launches if pmc_b else None
        ws = self.working_set_bytes()
        NX, NU = self.dims[0], self.dims[1]
        mode = int(kname.split('<')[1].split(',')[4])
        # resident workgroup slots of the kernel: the library's own occupancy query (workgroups of this kernel a CU holds x the CUs of THIS device)
        wg_per_cu, ncu, _ = prob.occupancy()
        check_every = prob.settings_dict().get('check_termination', 25) or 25
        slots = ncu * wg_per_cu
        # what the memory-side cache sees: the instances IN FLIGHT.  An instance re-reads its factor every iteration while it is resident -- for a whole
        # closed loop on the device-loop path, for a round of 25 iterations per launch on the stepwise one -- so the reuse distance of the stream is one
        # iteration of the resident instances, not the batch
        ws_active = self.working_set_bytes(slots)
        secs = admm_ms * 1e-3
        model_b = design_bytes / max(1, iters)                     # design bytes per ADMM iteration and QP, rounds and solves amortised
        common = {'kernel': kname, 'kernel_ms': admm_ms / launches, 'launches': res['launches'], 'steps_per_launch': res.get('chunk', 1),
                  'bytes_per_launch': design_bytes / launches, 'model_bytes_per_iter_per_qp': model_b, 'measured_bytes_per_iter_per_qp': pmc_b,
                  'traffic': traffic, 'traffic_GBps': (traffic * launches / secs / 1e9) if traffic else None,
                  'frac_counters': (traffic * launches / secs / HBM_PEAK) if traffic else None,
                  'traffic_over_model': (pmc_b / model_b) if pmc_b else None,
                  'traffic_profile': ('profiles/pmc_hbm_traffic.json:%s/%s (batch %s)' % (workload_key, path, pmc_batch if pmc_batch is not None else self.B)) if pmc_b else None,
                  'design_bytes_per_iter_per_qp': per_iter, 'design_bytes_per_round_per_qp': per_round, 'design_bytes_per_solve_per_qp': per_solve,
                  'working_set_bytes': ws, 'active_working_set_bytes': ws_active, 'fits_infinity_cache': bool(ws_active <= INFINITY_CACHE),
                  'resident_slots': slots, 'compute_units': ncu, 'check_every': check_every, 'shared_factor_fraction': shared_frac or None}
        if mode >= 100:
            # the register-resident backends (one workgroup per CU at a time): factor and iterate live in registers and LDS, an ADMM iteration reads
            # nothing from memory; what the kernel moves is per round (level fragments, owner values, check inputs) and per solve.  `frac` is those
            # design bytes (mpcqp_get_stream_bytes x the device-side counters of the timed region) / HIP-event kernel time / 8 TB/s -- the same
            # definition as for the bandwidth kernels -- and it is SMALL BY DESIGN: the kernel is bound by dependent level steps, not by HBM.
            # Named separately: mfma_issue_frac = executed v_mfma_f64_4x4x4 flops (512 per instruction, mpcqp_get_work) / time / the FP64 matrix
            # peak; useful_flop_frac = a quarter of that (a mat-vec uses one of the four B columns).
            mfma = prob.mfma_per_iter()
            flops = 512.0 * mfma * iters
            tail = self.tail_split(res, per_iter, per_round, per_solve, check_every) if path == 'device_loop' else None
            return dict(common, bound='hbm', achieved=achieved / 1e9, peak=HBM_PEAK / 1e9, unit='GB/s', frac=achieved / HBM_PEAK,
                        frac_excluding_tail=(tail['rate_before_tail'] / HBM_PEAK) if tail else None, tail=tail,
                        mfma_issue_frac=flops / secs / MFMA_F64_PEAK, useful_flop_frac=0.25 * flops / secs / MFMA_F64_PEAK,
                        mfma_flops_executed_per_launch=flops / launches, mfma_per_iter_per_qp=mfma, mfma_peak_TFLOPs=MFMA_F64_PEAK / 1e12)
        tail = self.tail_split(res, per_iter, per_round, per_solve, check_every) if path == 'device_loop' else None
        stream = MALL_READ_STREAM if ws_active <= INFINITY_CACHE else HBM_READ_STREAM
        return dict(common, bound='hbm', achieved=achieved / 1e9, peak=HBM_PEAK / 1e9, unit='GB/s', frac=achieved / HBM_PEAK,
                    frac_excluding_tail=(tail['rate_before_tail'] / HBM_PEAK) if tail else None, tail=tail,
                    frac_of_achievable=achieved / HBM_ACHIEVABLE, achievable_GBps=HBM_ACHIEVABLE / 1e9,      # (MI355X_MICROARCH.md: 6.29 TB/s measured with a float4 copy)
                    frac_of_read_stream=achieved / stream, read_stream_GBps=stream / 1e9, lds_state=lds_state,
                    algorithmic_8d={'bytes_per_launch': alg8d / launches, 'bytes_per_iter_per_qp': b_it8d, 'nnzL': prob.nnzL,
                                    'GBps': alg8d / secs / 1e9, 'frac_8d': alg8d / secs / HBM_PEAK})


# ----------------------------------------------------------------------------------------------------------------------
# the ONE line the driver parses: numbers only, a few kB (round 5's line carried every leg twice with its prose: 27.5 kB, and the
# driver recorded `parsed: null`).  Everything else goes to the side file bench_legs.json (LEGS_FILE below).
# ----------------------------------------------------------------------------------------------------------------------
LINE_BUDGET = 6000          # bytes; tests/test_bench_line.py holds the assembled line to it


def legs_path():
    """Where the full record of a run goes: $MPCQP_BENCH_LEGS, else gpurun_out/ if this tree has one (it travels back from the GPU box), else the repo root."""
    p = os.environ.get('MPCQP_BENCH_LEGS')
    if p:
        return p
    d = os.path.join(ROOT, 'gpurun_out')
    return os.path.join(d if os.path.isdir(d) else ROOT, 'bench_legs.json')


def _num(v, nd=6):
    """Numbers to `nd` significant digits (the line is read by a parser, not by the eye; the side file keeps every digit)."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        if v != v or v in (float('inf'), float('-inf')):
            return None
        return float('%.*g' % (nd, v))
    return v


def _pick(d, keys, nd=6):
    return {k: _num(d[k], nd) for k in keys if isinstance(d, dict) and k in d and d[k] is not None and not isinstance(d[k], (dict, list))}


ROOF_KEYS = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'kernel_ms', 'launches', 'steps_per_launch',
             'frac_counters', 'traffic_GBps', 'frac_excluding_tail', 'mfma_issue_frac', 'useful_flop_frac',
             'bytes_per_launch', 'model_bytes_per_iter_per_qp', 'measured_bytes_per_iter_per_qp', 'traffic_over_model', 'traffic_profile')


def compact_leg(leg, nd=5):
    """value + what bounds it, numbers only (kernel names and the rest: the side file)."""
    ro = leg.get('roofline') or {}
    sp = leg.get('launch_spread') or {}
    o = _pick(leg, ('batch', 'value', 'mean_admm_iters'), nd)
    o.update(_pick(ro, ('frac', 'frac_excluding_tail', 'frac_counters', 'kernel_ms', 'mfma_issue_frac'), 4))
    if sp.get('critical_path_ratio') is not None:
        o['critical_path_ratio'] = _num(sp['critical_path_ratio'], 4)
    return o


def compact_line(out):
    """The driver's line from the full record: every key of the bench contract, `roofline` and `cpu_baseline` as objects of numbers and short
    identifiers, one small object per leg.  No prose.  Pure function of `out` (tests/test_bench_line.py feeds it canned records)."""
    line = {k: _num(out.get(k)) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                          'vs_baseline', 'dtype', 'data', 'ranks_seen', 'devices_seen', 'collective_backend')}
    cfg = out.get('config') or {}
    line['config'] = _pick(cfg, ('workload', 'nx', 'nu', 'Np', 'n', 'm', 'batch_per_gpu', 'total_batch', 'eps_abs', 'eps_rel', 'path', 'parallelism'))
    line.update(_pick(out, ('mean_admm_iters', 'solved_fraction_last_step', 'refactorizations_per_solve')))
    line['roofline'] = _pick(out.get('roofline') or {}, ROOF_KEYS)
    cpu = out.get('cpu_baseline')
    if cpu:
        c = _pick(cpu, ('value', 'unit', 'cores', 'kind'))
        c['sample'] = str(cpu.get('sample', ''))[:160]
        ac = cpu.get('all_cores') or {}
        if 'value' in ac:
            c['all_cores'] = _pick(ac, ('value', 'cores'))
        line['cpu_baseline'] = c
    ue = out.get('u_err')
    if ue:
        line['u_err'] = {k: _pick(v, ('max_abs', 'max_rel', 'instances'), 4) for k, v in ue.items() if isinstance(v, dict)}
        line['u_err']['tolerance_rel'] = ue.get('north_star_tolerance_rel', 1e-6)
    line['per_rank'] = [_pick(r, ('rank', 'instances', 'first_instance', 'value', 'ms_per_step', 'roofline_frac', 'kernel_ms', 'scatter_ms', 'gather_calls', 'gather_ms_per_call', 'instances_sharing_factor'), 5)
                        for r in (out.get('per_rank') or [])]
    line['cold'] = _pick(out.get('cold') or {}, ('setup_ms', 'first_solve_ms', 'setup_ms_repeat', 'first_solve_ms_repeat', 'iters_per_instance'), 4)
    legs = {}
    if out.get('other_path'):
        legs[out['other_path']['path']] = _pick(out['other_path'], ('value', 'ms_per_step', 'mean_admm_iters'))
    if out.get('parity_setting'):
        legs['eps_1e-9'] = _pick(out['parity_setting'], ('value', 'ms_per_step', 'mean_admm_iters', 'solved_fraction_last_step'))
    if out.get('strong_scaling'):
        legs['strong_total1024'] = _pick(out['strong_scaling'], ('value', 'ms_per_step', 'batch_per_gpu', 'mean_admm_iters'))
    if out.get('hbm_leg'):
        legs['sweeps_b%d' % out['hbm_leg']['batch']] = compact_leg(out['hbm_leg'])
        if out['hbm_leg'].get('stepwise'):
            legs['sweeps_b%d_stepwise' % out['hbm_leg']['batch']] = compact_leg(out['hbm_leg']['stepwise'])
    if out.get('cfg5_leg'):
        c5 = compact_leg(out['cfg5_leg'])
        c5['eps_1e-9_value'] = _num((out['cfg5_leg'].get('parity_setting') or {}).get('value'))
        u5 = out['cfg5_leg'].get('u_err') or {}
        if 'eps_1e-09' in u5:
            c5['u_err_rel_eps_1e-9'] = _num(u5['eps_1e-09'].get('max_rel'), 4)
        legs['cfg5_b%d' % out['cfg5_leg']['batch']] = c5
        db = out['cfg5_leg'].get('double_batch') or {}
        if 'value' in db:
            legs['cfg5_b%d' % db['batch']] = compact_leg(db)
    for k, v in (out.get('small_batch_legs') or {}).items():
        if 'error' not in v:
            legs[k] = compact_leg(v)
    for k, v in (out.get('shared_model_legs') or {}).items():
        if 'error' not in v:
            legs[k] = _pick(v, ('batch', 'value', 'own_factor_value', 'register_resident_value', 'bit_identical_to_own_factor', 'mean_admm_iters', 'useful_flop_frac'), 5)
    pr = out.get('strong_scaling_projection')
    if pr:
        legs['projection_8gpu'] = _pick(pr, ('batch_per_gpu_at_8', 'projected_8gpu_value', 'projected_vs_1gpu_full_batch', 'weak_scaling_projection_8gpu_value'))
    lat = out.get('latency') or {}
    for k, v in lat.items():
        if isinstance(v, dict) and 'update_us_median' in v:
            legs['latency_' + k] = _pick(v, ('update_us_median', 'update_us_p95', 'kernel_us', 'raw_c_abi_step_us', 'cpu_oracle_update_us_median', 'mean_admm_iters'), 4)
    line['legs'] = legs
    ro = out.get('real_osqp') or {}
    line['real_osqp'] = _pick(ro, ('osqp_available', 'osqp_version', 'all_equal'))
    line['legs_file'] = out.get('legs_file')
    # the budget is a hard one: shed detail (never the contract's keys) until the line fits
    def fits():
        return len(json.dumps(line)) <= LINE_BUDGET
    if not fits():
        line['legs'] = {k: _pick(v, ('batch', 'value', 'frac', 'update_us_median', 'projected_8gpu_value'), 4) for k, v in legs.items()}
    for drop in ('cold', 'real_osqp', 'legs'):
        if fits():
            break
        line.pop(drop, None)
    return line


def emit(out):
    """Side file first (the whole record), then the compact line as the LAST line on stdout."""
    path = legs_path()
    out['legs_file'] = os.path.relpath(path, ROOT)
    try:
        with open(path, 'w') as f:
            json.dump(out, f, default=lambda o: o.tolist() if hasattr(o, 'tolist') else repr(o))
    except OSError as e:
        out['legs_file'] = 'unwritable: %r' % (e,)
    sys.stdout.flush()
    print(json.dumps(compact_line(out)), flush=True)


# ----------------------------------------------------------------------------------------------------------------------
# single-controller latency legs (BASELINE configs[1] and the reference notebook's shape)
# ----------------------------------------------------------------------------------------------------------------------
def shared_model_leg(args, dims, B, torch, dev, steps=40, warmup=20, chunk=20):
    """One model, many states (test_scripts/example_mpc_function.py:105-111, SURVEY 8(e) last paragraph): B instances of ONE random model of the workload's shape,
    set up from one common state (what one reference controller's setup() is), then scattered states and per-instance noise in the closed device loop.  The
    bandwidth kernel with every instance streaming its own factor, the same with mpcqp_share_factor (one copy, out of L2), and the register-resident kernel."""
    from pympc_amd.solver import BatchProblem
    from pympc_amd import fixtures, _lib
    NX, NU, NP, XBOX = dims
    f64 = torch.float64
    kw = fixtures.random_lti(0, nx=NX, nu=NU, Np=NP, xbox=XBOX)
    Ad, Bd, x_common = (np.asarray(kw[k], dtype=float) for k in ('Ad', 'Bd', 'x0'))
    rng = np.random.default_rng(1)
    X0 = x_common[None, :] * rng.uniform(0.2, 1.0, size=(B, 1)) * rng.choice([-1.0, 1.0], size=(B, NX))
    W = torch.from_numpy(0.01 * rng.standard_normal((warmup + steps, B, NX))).to(dev)
    ones = lambda k, v: np.full((B, k), v)
    outs = (torch.empty((chunk + 1, B, NX), dtype=f64, device=dev), torch.empty((chunk, B, NU), dtype=f64, device=dev),
            torch.empty((chunk, B), dtype=torch.int32, device=dev), torch.empty((chunk, B), dtype=torch.int32, device=dev))

    def run(backend, share):
        prob = BatchProblem(B, NX, NU, NP, device=dev.index, stream=torch.cuda.current_stream(dev).cuda_stream, eps_abs=args.eps, eps_rel=args.eps, warm_start=1, backend=backend,
                            tuning=0 if share else _lib.TUNE_NO_SHARE)      # (setup shares by itself unless told not to)
        prob.setup(Ad, Bd, np.eye(NX), np.eye(NX), 0.1 * np.eye(NU), 0.1 * np.eye(NU), ones(NX, -XBOX), ones(NX, XBOX), ones(NU, -1.0), ones(NU, 1.0),
                   ones(NU, -0.5), ones(NU, 0.5), ones(NU, 0.0), np.full((B, 1), 1e6), np.broadcast_to(x_common, (B, NX)), ones(NU, 0.0), np.zeros((B, NX)))
        prob.solve_async(); prob.synchronize()
        nshared = prob.share_factor() if share else 0
        prob.update(X0, np.zeros((B, NU)))
        hist = []
        for c in range(0, warmup, chunk):
            prob.mpc_run(chunk, w=W[c:c + chunk], out=outs); hist.append(outs[1].clone())
        torch.cuda.synchronize()
        prob.stats(reset=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for c in range(warmup, warmup + steps, chunk):
            prob.mpc_run(chunk, w=W[c:c + chunk], out=outs); hist.append(outs[1].clone())
        e1.record(); torch.cuda.synchronize()
        secs = 1e-3 * e0.elapsed_time(e1)
        st = prob.stats(reset=True)
        flops = 512.0 * prob.mfma_per_iter() * st[0]
        r = {'batch': B, 'backend': backend, 'shared_factor': bool(share), 'instances_sharing': nshared, 'value': B * steps / secs, 'unit': 'QP-solves/s', 'ms_per_step': 1e3 * secs / steps,
             'mean_admm_iters': st[0] / max(1, st[3]), 'refactorizations': st[2], 'kernel': prob.kernel_name(True), 'solved_fraction_last_step': float((outs[2][-1] == 1).double().mean()),
             'mfma_issue_frac': flops / secs / MFMA_F64_PEAK, 'useful_flop_frac': 0.25 * flops / secs / MFMA_F64_PEAK}
        prob.close()
        return r, torch.cat(hist)

    own, U0 = run('sweeps', False)
    shr, U1 = run('sweeps', True)
    reg, U2 = run('auto', False)
    return dict(shr, own_factor_value=own['value'], speedup_over_own_factor=shr['value'] / own['value'], bit_identical_to_own_factor=bool(torch.equal(U0, U1)),
                register_resident_value=reg['value'], register_resident_kernel=reg['kernel'],
                max_abs_input_difference_to_register_resident=float((U1 - U2).abs().max()),
                definition='device loop, %d timed steps in launches of %d after %d warm-up steps; value = the bandwidth kernel with ONE shared factor (mpcqp_share_factor)' % (steps, chunk, warmup))


def latency_leg(kind, nsim=300):
    """One controller through the drop-in class, `MPCController.update()` per step (pyMPC/mpc.py:338-364), median microseconds;
    beside it the CPU oracle on the same loop.  kind = 'cfg2': examples/example_inverted_pendulum.py:10-69 (4,1,20);
    'notebook': examples/example_inverted_pendulum_kalman.ipynb cells 3/12/13/15 (4,1,Np=150,Nc=75; the notebook's own
    timing cell reports about 1.05-1.2 ms per step for its OSQP-backed controller); 'kalman_np200': the script version of that example,
    examples/example_inverted_pendulum_kalman.py:19,71-110 (Ts = 5 ms, Np = Nc = 200, eps_feas = 1e3)."""
    import warnings
    from pympc_amd import MPCController, fixtures
    kw = fixtures.cart_pole()
    if kind == 'notebook':
        kw.update(Np=150, Nc=75)
    if kind == 'kalman_np200':                       # examples/example_inverted_pendulum_kalman.py:19,71-110: Ts = 5 ms, Np = Nc = 200, eps_feas = 1e3
        kw = fixtures.cart_pole_kalman()

    def loop(K):
        x = np.array(kw['x0'], dtype=float)
        ts, its = [], []
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            K.setup()
            for _ in range(nsim):
                u = K.output()
                x = kw['Ad'] @ x + kw['Bd'] @ u
                t = time.perf_counter(); K.update(x); ts.append(time.perf_counter() - t)
                its.append(K.res.info.iter)
        return 1e6 * np.array(ts), np.array(its)

    K = MPCController(**kw)
    ts, its = loop(K)
    bp = K.prob.batch_problem
    bp.profile(enable=True, reset=True)
    x = np.array(K.x0_rh, dtype=float)
    t0 = time.perf_counter()
    for _ in range(200):
        bp.step_host(x0=x[None, :])                  # what the class calls: mpcqp_step_host (one launch, mapped host memory, no copies)
    raw_us = 1e6 * (time.perf_counter() - t0) / 200
    ms, nl = bp.profile(enable=False)
    t0 = time.perf_counter()
    for _ in range(200):
        bp.update(x0=x[None, :]); bp.solve_async(); bp.u0()
    raw3_us = 1e6 * (time.perf_counter() - t0) / 200
    out = dict(workload='%s: one cart-pole MPCController, nx=4 nu=1 Np=%d Nc=%d' % (kind, kw['Np'], kw.get('Nc', kw['Np'])),
               update_us_median=float(np.median(ts)), update_us_p95=float(np.percentile(ts, 95)), mean_admm_iters=float(its.mean()),
               raw_c_abi_step_us=raw_us, raw_c_abi_update_solve_u0_us=raw3_us, kernel_us=1e3 * ms / max(1, nl), kernel=bp.kernel_name(loop=False))
    from oracle.osqp_oracle import OSQP
    Ko = MPCController(**kw); Ko.prob = OSQP()
    tso, itso = loop(Ko)
    out['cpu_oracle_update_us_median'] = float(np.median(tso))
    # where the tail comes from: update() time follows the ADMM iterations of the step (OSQP checks every 25) -- the first steps after the cold
    # start need several rounds and a rho update (one refactorization) -- on the CPU in the same proportion
    out['cpu_oracle_update_us_p95'] = float(np.percentile(tso, 95))
    out['admm_iters_median'], out['admm_iters_p95'] = float(np.median(its)), float(np.percentile(its, 95))
    out['us_per_admm_iter_median'] = float(np.median(ts / np.maximum(1, its)))
    out['tail_note'] = 'p95 / median of update() = %.2f on the GPU, %.2f on the CPU oracle, %.2f in ADMM iterations' % (
        np.percentile(ts, 95) / np.median(ts), np.percentile(tso, 95) / np.median(tso), np.percentile(its, 95) / max(1.0, np.median(its)))
    if kind == 'notebook':
        out['reference_note'] = 'examples/example_inverted_pendulum_kalman.ipynb cell 17: about 1.05-1.2 ms per MPC step (OSQP, the author\'s laptop)'
    return out


def dry_run(args, rank, world, dev, seen, torch, dist):
    """The distributed plumbing of the bench without the solver: the packed scatter from rank 0 (two arrays, one collective), a stand-in
    per-shard result, the all-gather of "u*" per step and per launch, a per-rank report, one JSON line.  `--total-batch T` as in the real
    run: T need not divide by the ranks (the last one is short).  Used by the CPU tests of `bench.py --gpus N` (gloo) -- nothing here is a
    measurement."""
    from pympc_amd import sharding
    total = args.total_batch if args.total_batch is not None else 4 * world
    lo, hi = sharding.shard_range(total, rank, world)
    full = None
    if rank == 0:
        full = {'x0': torch.arange(total * 3, dtype=torch.float64, device=dev).reshape(total, 3), 'Ad': torch.arange(total * 4, dtype=torch.float64, device=dev).reshape(total, 2, 2)}
    if args.shared_model:      # ONE model broadcast, the states alone scattered (what Shard does with --shared-model); the stand-in model is instance 0's
        mdl = sharding.broadcast_model({'Ad': full['Ad'][0]} if rank == 0 else None, {'Ad': (2, 2)}, dev)
        loc = sharding.scatter_instances({'x0': full['x0']} if rank == 0 else None, {'x0': (3,)}, None, dev, total=total)
        loc['Ad'] = mdl['Ad'].expand(hi - lo, 2, 2).contiguous()
    else:
        loc = sharding.scatter_instances(full, {'Ad': (2, 2), 'x0': (3,)}, None, dev, total=total)
    assert loc['x0'].shape == (hi - lo, 3) and loc['Ad'].shape == (hi - lo, 2, 2)
    u = 2.0 * loc['x0'][:, :2] + loc['Ad'][:, 1, 1:2]
    u_all = sharding.gather_inputs(u, total=total)
    traj = torch.stack([u + 10.0 * k for k in range(3)])              # [steps, count, nu]: a device-loop launch's inputs
    tr_all = sharding.gather_trajectory(traj, total=total)
    mine = {'rank': rank, 'instances': hi - lo, 'first_instance': lo}
    per_rank = [mine]
    if comm_on(world):
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        dist.barrier()
    if rank == 0:
        x0 = torch.arange(total * 3, dtype=torch.float64).reshape(total, 3)
        Ad_all = torch.arange(total * 4, dtype=torch.float64).reshape(total, 2, 2)
        expect = 2.0 * x0[:, :2] + (Ad_all[:1].expand(total, 2, 2) if args.shared_model else Ad_all)[:, 1, 1:2]
        ok = bool(torch.equal(u_all.cpu(), expect)) and tuple(tr_all.shape) == (3, total, 2) and bool(torch.equal(tr_all[2].cpu(), expect + 20.0))
        print(json.dumps(dict(dry_run=True, n_gpus=world, total_batch=total, gathered_ok=ok, shared_model=bool(args.shared_model), per_rank=per_rank, **seen)))
    if comm_on(world):
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100, help='timed MPC steps (SURVEY 8d cfg-3: 100-step receding horizon)')
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=None, help='instances per GPU (weak scaling; default 1024, cfg5: 512)')
    ap.add_argument('--total-batch', type=int, default=None, help='instances in total, split evenly over the GPUs (strong scaling)')
    ap.add_argument('--eps', type=float, default=1e-3)
    ap.add_argument('--chunk', type=int, default=None, help='device loop: steps per kernel launch (default: the timed steps in equal launches of at most 50)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--shared-model', action='store_true', help='every instance (on every rank) is a copy of ONE controller -- the first instance\'s model, broadcast once -- and only the states are scattered '
                                                                '(test_scripts/example_mpc_function.py:105-111, SURVEY 8e last paragraph); implies --no-cpu-baseline')
    ap.add_argument('--no-shared-model-leg', action='store_true', help='skip the one-model-many-states legs (mpcqp_share_factor)')
    ap.add_argument('--no-other-path', action='store_true', help='skip the secondary measurements (other path, parity setting, strong-scaling / HBM / latency legs)')
    ap.add_argument('--no-refactor-timing', action='store_true', help='skip the timing of the factorization alone (profiling runs: its launches carry the solve kernel\'s name)')
    ap.add_argument('--path', default='device_loop', choices=['stepwise', 'device_loop'],
                    help='stepwise: update()/solve()/output() per step from the host (the reference call pattern); '
                         'device_loop: the same K steps inside mpcqp_mpc_loop (SURVEY 8f-1)')
    ap.add_argument('--workload', default='cfg3', choices=['cfg3', 'cfg5', 'cfg2', 'notebook', 'kalman_np200'],
                    help='cfg3: 1024 x (12,4,30) (headline); cfg5: 512 x (20,8,100), tight state box (SURVEY 8d); '
                         'cfg2 / notebook: single-controller latency legs only')
    ap.add_argument('--hbm-leg-batch', type=int, default=4096, help='cfg3, 1 GPU: second leg with a working set beyond the Infinity Cache (0 = skip)')
    ap.add_argument('--cfg5-leg-batch', type=int, default=512, help='cfg3, 1 GPU: BASELINE configs[4] (512 x (20,8,100)) as a leg of the default line (0 = skip)')
    ap.add_argument('--backend', default=None, choices=['sweeps', 'dense', 'bcr8'],
                    help='force a KKT backend for the headline shard (mpcqp_settings.backend; default: the library chooses) -- profiling runs of the bandwidth kernel at the headline batch')
    ap.add_argument('--tuning', type=int, default=None, help='mpcqp_settings.tuning for the headline shard (development: switch a mechanism off, see enum mpcqp_tuning)')
    ap.add_argument('--dry-run', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.shared_model:
        args.no_cpu_baseline = args.no_other_path = True      # (the CPU legs and the u* references rebuild instance i's OWN model from its seed)
    self_launch_if_needed(args)

    import torch
    import torch.distributed as dist
    rank, world, local_rank, dev, seen = init_distributed(args, torch, dist)
    if args.dry_run:
        return dry_run(args, rank, world, dev, seen, torch, dist)
    if args.workload in ('cfg2', 'notebook', 'kalman_np200'):
        if rank == 0:
            leg = latency_leg(args.workload)
            print(json.dumps({'metric': 'MPCController.update() latency, one controller (BASELINE configs[1])', 'value': leg['update_us_median'], 'unit': 'us',
                              'n_gpus': 1, 'higher_is_better': False, 'dtype': 'f64', 'data': 'synthetic', 'config': {'workload': leg['workload']}, 'latency': leg}))
        if comm_on(world):
            dist.barrier(); dist.destroy_process_group()
        return
    dims = WORKLOADS[args.workload][:4]
    NX, NU, NP, XBOX = dims
    if args.total_batch is not None:
        TOTAL, scaling = args.total_batch, 'strong'                 # (need not divide: the last rank is short, pympc_amd/sharding.py)
    else:
        TOTAL, scaling = (args.batch if args.batch is not None else WORKLOADS[args.workload][4]) * world, 'weak'

    from pympc_amd.solver import forced_settings
    with forced_settings(**dict(({'backend': args.backend} if args.backend else {}), **({'tuning': args.tuning} if args.tuning is not None else {}))):
        sh = Shard(args, dims, None, rank, world, dev, 0, torch, dist, total=TOTAL)
    B = sh.B                                                          # this rank's instances
    prob = sh.prob
    res = sh.measure(args.path, args.steps, args.warmup)
    elapsed, iters, refacts, solves = res['elapsed'], res['iters'], res['refacts'], res['solves']
    infos = prob.infos()
    n_solved = sum(1 for i in infos if i.status == 1)
    samples = [] if args.shared_model else [dict(sh.sample_point(), eps=args.eps)]
    other = None
    if not args.no_other_path:
        oname = 'stepwise' if args.path == 'device_loop' else 'device_loop'
        o = sh.measure(oname, args.steps, args.warmup)
        other = {'path': oname, 'value': TOTAL * args.steps / o['elapsed'], 'ms_per_step': 1e3 * o['elapsed'] / args.steps,
                 'mean_admm_iters': o['iters'] / max(1, o['solves'])}
    parity = None
    if not args.no_other_path and args.eps > 1e-8:
        # the same loop at the parity setting eps = 1e-9 (the tolerance the u* comparison is made at)
        prob.update_settings(eps_abs=1e-9, eps_rel=1e-9)
        pr = sh.measure(args.path, args.steps, args.warmup)
        pinf = prob.infos()
        parity = {'eps_abs': 1e-9, 'eps_rel': 1e-9, 'path': args.path, 'value': TOTAL * args.steps / pr['elapsed'],
                  'ms_per_step': 1e3 * pr['elapsed'] / args.steps, 'mean_admm_iters': pr['iters'] / max(1, pr['solves']),
                  'solved_fraction_last_step': sum(1 for i in pinf if i.status == 1) / B}
        samples.append(dict(sh.sample_point(), eps=1e-9))
        prob.update_settings(eps_abs=args.eps, eps_rel=args.eps)
    # what one rho update costs: the block factorization of every instance, timed alone (mpcqp_refactor rewrites the factor
    # that is already in place); the steady-state loop above needs none, the cold solve a few per instance
    refactor_ms = None if args.no_refactor_timing else sh.refactor_ms()
    roof_local = sh.roofline(res, args.path, args.workload + ('_' + args.backend if args.backend else '') + ('_b%d' % B if args.workload == 'cfg3' and B != WORKLOADS['cfg3'][4] else ''))
    roof = roof_local if rank == 0 else None
    # what every rank did, beside the job's line: its share, its own rate (its clock, before the max over ranks), its kernel's roofline fraction,
    # and what the two exchanges cost it -- the scatter of the problem data (one packed collective at setup) and the all-gathers of u*
    mine = {'rank': rank, 'instances': B, 'first_instance': sh.first_local, 'value': B * args.steps / res['elapsed_local'], 'ms_per_step': 1e3 * res['elapsed_local'] / args.steps,
            'roofline_bound': roof_local['bound'], 'roofline_frac': roof_local['frac'], 'kernel': roof_local['kernel'], 'kernel_ms': roof_local['kernel_ms'],
            'instances_sharing_factor': sh.instances_sharing, 'scatter_ms': sh.scatter_ms, 'gather_calls': sh.gathers, 'gather_ms_per_call': (sh.gather_ms / sh.gathers) if sh.gathers else None}
    per_rank = [mine]
    if comm_on(world):
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    totals, cold, n, m = sh.totals, sh.cold, prob.n, prob.m
    knames = {k: prob.kernel_name(loop=(k == 'loop')) for k in totals}

    # ---- secondary legs (each on a problem set of its own; the headline shard is released first)
    extra, cfg5_samples = {}, None
    if not args.no_other_path:
        del sh, prob
        torch.cuda.empty_cache()
        if world > 1 and scaling == 'weak' and args.workload == 'cfg3':
            # BASELINE configs[3] read literally: the SAME 1024 instances split over the N GPUs
            s2 = Shard(args, dims, None, rank, world, dev, 0, torch, dist, total=1024)
            r2 = s2.measure(args.path, args.steps, args.warmup)
            extra['strong_scaling'] = {'scaling': 'strong', 'total_batch': 1024, 'batch_per_gpu': 1024 // world, 'path': args.path,
                                       'value': 1024 * args.steps / r2['elapsed'], 'ms_per_step': 1e3 * r2['elapsed'] / args.steps,
                                       'mean_admm_iters': r2['iters'] / max(1, r2['solves'])}
            del s2
            torch.cuda.empty_cache()
        if world == 1 and args.workload == 'cfg3' and args.hbm_leg_batch and args.hbm_leg_batch != B:
            # the BANDWIDTH kernel (forced: since the second half of round 5 the library runs this shape on the register-resident kernel at every batch size) with
            # finished slots refilling; counter traffic from the profile of this very command shape (profiles/pmc_hbm_traffic.json, key cfg3_sweeps_b<batch>)
            with forced_settings(backend='sweeps'):
                s3 = Shard(args, dims, args.hbm_leg_batch, rank, world, dev, 0, torch, dist)
            r3 = s3.measure(args.path, min(args.steps, 25), min(args.warmup, 25) or 5)
            ro3 = s3.roofline(r3, args.path, 'cfg3_sweeps_b%d' % args.hbm_leg_batch)
            # the stepwise API at this batch: every solve walks all the instances (878 MB) -- but 25 iterations at a time, and an instance re-reads its
            # factor 25 times while it is resident: HBM delivers the first touch, the Infinity Cache the other 24 (active set: 1024 resident instances, 219 MB)
            r3s = s3.measure('stepwise', min(args.steps, 25), min(args.warmup, 25) or 5)
            ro3s = s3.roofline(r3s, 'stepwise', None)
            extra['hbm_leg'] = {'batch': args.hbm_leg_batch, 'value': args.hbm_leg_batch * min(args.steps, 25) / r3['elapsed'], 'unit': 'QP-solves/s',
                                'ms_per_step': 1e3 * r3['elapsed'] / min(args.steps, 25), 'mean_admm_iters': r3['iters'] / max(1, r3['solves']),
                                'launch_spread': r3.get('launch_spread'),
                                'stepwise': {'value': args.hbm_leg_batch * min(args.steps, 25) / r3s['elapsed'], 'unit': 'QP-solves/s', 'ms_per_step': 1e3 * r3s['elapsed'] / min(args.steps, 25),
                                             'mean_admm_iters': r3s['iters'] / max(1, r3s['solves']),
                                             'roofline': ro3s},
                                'roofline': ro3}
            # (4096 instances pass through the resident workgroup slots; an instance re-reads its factor every ADMM iteration while it is resident, so the stream re-reads an
            #  ACTIVE set of 219 MB whatever the batch -- inside the 256 MiB Infinity Cache, like the headline; what the larger batch removes is the idle tail.  `stepwise`
            #  walks all 878 MB per solve, 25 iterations per launch: HBM serves the first of them.  The leg whose active set (768 MB) IS beyond the Infinity Cache is cfg5_leg.)
            del s3
            torch.cuda.empty_cache()
        if world == 1 and args.workload == 'cfg3' and args.cfg5_leg_batch:
            # BASELINE configs[4] in the default line: 512 x (20,8,100), tight state box (slack rows active), 50 warm-started closed-loop
            # steps after 25 warm-up steps, at the reference's default tolerance and at the parity setting
            d5 = WORKLOADS['cfg5'][:4]
            B5 = args.cfg5_leg_batch
            s5 = Shard(args, d5, B5, rank, world, dev, 0, torch, dist)
            r5 = s5.measure('device_loop', 50, 25)
            ro5 = s5.roofline(r5, 'device_loop', 'cfg5')
            inf5 = s5.prob.infos()
            cfg5_samples = [dict(s5.sample_point(16), eps=args.eps)]
            s5.prob.update_settings(eps_abs=1e-9, eps_rel=1e-9)
            p5 = s5.measure('device_loop', 50, 25)
            cfg5_samples.append(dict(s5.sample_point(16), eps=1e-9))
            extra['cfg5_leg'] = {'workload': 'cfg-5: %d random stable LTI MPC instances (nx=%d, nu=%d, Np=Nc=%d, n=%d, m=%d), state box +-%g (slack active), '
                                             'warm-started receding horizon' % (B5, d5[0], d5[1], d5[2], s5.prob.n, s5.prob.m, d5[3]),
                                 'batch': B5, 'steps': 50, 'warmup': 25, 'eps_abs': args.eps, 'eps_rel': args.eps, 'path': 'device_loop',
                                 'value': B5 * 50 / r5['elapsed'], 'unit': 'QP-solves/s', 'ms_per_step': 1e3 * r5['elapsed'] / 50,
                                 'mean_admm_iters': r5['iters'] / max(1, r5['solves']), 'solved_fraction_last_step': sum(1 for i in inf5 if i.status == 1) / B5,
                                 'launch_spread': r5.get('launch_spread'), 'cold': s5.cold,
                                 'roofline': ro5,
                                 'parity_setting': {'eps_abs': 1e-9, 'eps_rel': 1e-9, 'value': B5 * 50 / p5['elapsed'], 'ms_per_step': 1e3 * p5['elapsed'] / 50,
                                                    'mean_admm_iters': p5['iters'] / max(1, p5['solves']), 'launch_spread': p5.get('launch_spread')}}
            del s5
            torch.cuda.empty_cache()
            # ... and twice that batch: beyond the 512 resident slots of this kernel a closed-loop launch is persistent (workgroups take (instance, step
            # range) items off a queue) and the stragglers' tail is filled by other instances -- what the 512-batch cannot have (all of it is resident at once)
            try:
                s5b = Shard(args, d5, 2 * B5, rank, world, dev, 0, torch, dist)
                r5b = s5b.measure('device_loop', 50, 25)
                extra['cfg5_leg']['double_batch'] = {'batch': 2 * B5, 'value': 2 * B5 * 50 / r5b['elapsed'], 'ms_per_step': 1e3 * r5b['elapsed'] / 50, 'mean_admm_iters': r5b['iters'] / max(1, r5b['solves']),
                                                     'launch_spread': r5b.get('launch_spread'), 'roofline': s5b.roofline(r5b, 'device_loop', None)}
                del s5b
            except Exception as e:
                extra['cfg5_leg']['double_batch'] = {'error': repr(e)}
            torch.cuda.empty_cache()
        if world == 1 and args.workload == 'cfg3' and scaling == 'weak' and B >= 8 and B % 8 == 0 and args.path == 'device_loop':
            # BASELINE configs[3] read literally (the SAME batch split over 8 GPUs): what ONE GPU does with its eighth, measured here --
            # the 8-GPU figure this projects to has no data-path collective to lose anything in (u* all-gather: 32 B per instance and step)
            s8 = Shard(args, dims, B // 8, rank, world, dev, 0, torch, dist)
            r8 = s8.measure(args.path, args.steps, args.warmup)
            v8 = (B // 8) * args.steps / r8['elapsed']
            ro8 = s8.roofline(r8, args.path, 'cfg3_b%d' % (B // 8))
            extra['small_batch_legs'] = {'b%d' % (B // 8): {'batch': B // 8, 'value': v8, 'ms_per_step': 1e3 * r8['elapsed'] / args.steps, 'launch_spread': r8.get('launch_spread'), 'roofline': ro8}}
            if B % 4 == 0:
                s4 = Shard(args, dims, B // 4, rank, world, dev, 0, torch, dist)
                r4 = s4.measure(args.path, args.steps, args.warmup)
                extra['small_batch_legs']['b%d' % (B // 4)] = {'batch': B // 4, 'value': (B // 4) * args.steps / r4['elapsed'], 'ms_per_step': 1e3 * r4['elapsed'] / args.steps,
                                                               'launch_spread': r4.get('launch_spread'), 'roofline': s4.roofline(r4, args.path, 'cfg3_b%d' % (B // 4))}
                del s4
                torch.cuda.empty_cache()
            extra['strong_scaling_projection'] = {'total_batch': B, 'batch_per_gpu_at_8': B // 8, 'measured_1gpu_value_at_that_batch': v8, 'kernel': s8.prob.kernel_name(loop=True),
                                                  'projected_8gpu_value': 8 * v8, 'projected_vs_1gpu_full_batch': 8 * v8 / (TOTAL * args.steps / elapsed),
                                                  'weak_scaling_projection_8gpu_value': 8 * TOTAL * args.steps / elapsed}
            # (strong scaling -- total batch fixed -- leaves B/8 instances per GPU: one 512-thread workgroup per instance, half the CUs idle at 128, and a launch as long as its
            #  slowest instance; weak scaling -- B per GPU -- is what --gpus N measures, --total-batch the strong reading on real GPUs)
            del s8
            torch.cuda.empty_cache()
            # the BANDWIDTH kernel forced at the headline batch: the headline kernel of rounds 1-4 and of the first half of round 5 (HBM / Infinity-Cache bound: its
            # own roofline, counter traffic from profiles key cfg3_sweeps); the library now runs this shape on the register-resident kernel at every batch size
            try:
                with forced_settings(backend='sweeps'):
                    sL = Shard(args, dims, B, rank, world, dev, 0, torch, dist)
                rL = sL.measure(args.path, args.steps, args.warmup)
                extra['small_batch_legs']['bandwidth_kernel_b%d' % B] = {'batch': B, 'value': B * args.steps / rL['elapsed'], 'ms_per_step': 1e3 * rL['elapsed'] / args.steps,
                                                                         'launch_spread': rL.get('launch_spread'), 'roofline': sL.roofline(rL, args.path, 'cfg3_sweeps')}
                del sL
            except Exception as e:
                extra['small_batch_legs']['bandwidth_kernel_b%d' % B] = {'error': repr(e)}
            torch.cuda.empty_cache()
        if world == 1 and args.workload == 'cfg3' and args.path == 'device_loop' and not args.no_shared_model_leg:
            extra['shared_model_legs'] = {}
            for Bs in (B, 4 * B):
                try:
                    extra['shared_model_legs']['shared_model_b%d' % Bs] = shared_model_leg(args, dims, Bs, torch, dev)
                except Exception as e:
                    extra['shared_model_legs']['shared_model_b%d' % Bs] = {'error': repr(e)}
                torch.cuda.empty_cache()
        if rank == 0 and args.workload == 'cfg3':
            try:
                extra['latency'] = {'cfg2': latency_leg('cfg2', nsim=200), 'notebook': latency_leg('notebook', nsim=100), 'kalman_np200': latency_leg('kalman_np200', nsim=60)}
            except Exception as e:
                extra['latency'] = {'error': repr(e)}

    if rank == 0:
        out = {
            'metric': 'QP-solves/sec (MPC steps/sec) at nx=%d nu=%d Np=%d; max |u*-u*_ref|' % (NX, NU, NP),
            'value': TOTAL * args.steps / elapsed,
            'unit': 'QP-solves/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'ranks_seen': seen['ranks_seen'], 'devices_seen': seen['devices_seen'], 'collective_backend': seen['backend'],
            'config': {'workload': ('%s: %d copies per GPU of ONE (nx=%d, nu=%d, Np=Nc=%d) random stable LTI MPC controller, states scattered, warm-started receding horizon' if args.shared_model else
                                    '%s: %d x (nx=%d, nu=%d, Np=Nc=%d) random stable LTI MPC instances per GPU, warm-started receding horizon')
                                   % ('cfg-3' if args.workload == 'cfg3' else 'cfg-5', B, NX, NU, NP),
                       'nx': NX, 'nu': NU, 'Np': NP, 'n': n, 'm': m,
                       'batch_per_gpu': B, 'total_batch': TOTAL, 'eps_abs': args.eps, 'eps_rel': args.eps, 'path': args.path,
                       'parallelism': 'instances sharded over %d GPU(s); RCCL scatter of data, all-gather of u*' % world},
            'mean_admm_iters': iters / max(1, solves),
            'solved_fraction_last_step': n_solved / B,
            'refactorizations_per_solve': refacts / max(1, solves),
            'refactorization': {'per_solve_timed_region': refacts / max(1, solves), 'ms_per_batch': refactor_ms, 'us_per_instance_amortised': (1e3 * refactor_ms / B) if refactor_ms else None,
                                'note': 'block LDL factorization of all %d instances in one launch (one rho update each); 0 per solve in the warm '
                                        'receding-horizon loop, a few per instance during the cold solve' % B},
            'cold': cold,
            'roofline': roof,
            'per_rank': per_rank,
            'launch_spread': res.get('launch_spread'),
            'accounting': {'timed': {'iters': iters, 'rounds': res['checks'], 'solves': solves, 'launches': res['launches'], 'kernel_ms_total': res['run_ms']},
                           'process_totals': {knames[k]: dict(iters=v[0], rounds=v[1], solves=v[2]) for k, v in totals.items()}},
            'other_path': other,
            'parity_setting': parity,
        }
        out.update(extra)
        cpu, refs = cpu_legs(dims, args.eps, samples, want_baseline=not args.no_cpu_baseline)
        if cpu:
            out['cpu_baseline'] = cpu
        def u_err_block(samples, refs):
            err = {}
            for s, ref in zip(samples, refs):
                ok = s['solved'] & np.isfinite(ref).all(axis=1)
                d = np.abs(s['u'][ok] - ref[ok]).max() if ok.any() else float('nan')
                sc = max(1e-3, np.abs(ref[ok]).max()) if ok.any() else 1.0
                err['eps_%g' % s['eps']] = {'max_abs': float(d), 'max_rel': float(d / sc), 'instances': int(ok.sum())}
            return dict(err, definition='max over the sample of |u* - u*_ref|_inf; rel = / max |u*_ref|_inf',
                        reference='oracle/osqp_ref.c at eps 1e-10 on the same (x0, u_-1): the QP the device solved in one more warm-started step',
                        north_star_tolerance_rel=1e-6)

        if refs:
            out['u_err'] = u_err_block(samples, refs)
        if cfg5_samples and 'cfg5_leg' in out:
            _, refs5 = cpu_legs(WORKLOADS['cfg5'][:4], args.eps, cfg5_samples, want_baseline=False)
            out['cfg5_leg']['u_err'] = u_err_block(cfg5_samples, refs5)
        out['real_osqp'] = real_osqp_pin() if not args.no_cpu_baseline else {'osqp_available': osqp_available()}
    if comm_on(world):
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        emit(out)                                                # the LAST thing this job writes to stdout


if __name__ == '__main__':
    main()
