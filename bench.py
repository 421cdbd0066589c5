#!/usr/bin/env python3
"""bench.py -- QP-solves/s of the MPC hot path on MI355X (BASELINE.json metric) and max |u* - u*_ref|.

One "step" = one closed-loop MPC step of the whole batch: plant update x+ = Ad x + Bd u* + w,
QP refresh (q, l, u from the new x0 and u_{-1}) and one warm-started ADMM solve of every instance, u* fetched.
Workload (BASELINE.json configs[2]): 1024 seed-pinned random stable LTI
systems nx=12, nu=4, Np=30 per GPU, reference-default tolerances (eps_abs=eps_rel=1e-3,
pyMPC/mpc.py:80), synthetic data, FP64, all inputs resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 100 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --workload cfg5                # BASELINE configs[4]: 512 x (20,8,100), Delta-u + slack rows active
    ... bench.py --gpus N --total-batch 1024       # BASELINE configs[3] read as strong scaling: the SAME 1024 instances over N GPUs

Two ways through the same library, both measured, `--path` chooses which one is `value` (the other is `other_path`):
  device_loop (default): the K steps run inside mpcqp_mpc_loop launches (output -> plant -> update -> solve per
                         instance on the device), at most 25 steps per launch (--chunk);
  stepwise             : the reference's call pattern, update()/solve()/output() per step from the host.
Both give bit-identical trajectories (tests/test_gpu_parity.py::test_device_loop_*).

Rank 0 prints ONE JSON line.  With N > 1 the instances are sharded over ranks (weak scaling: `--batch` per GPU; strong
scaling with --total-batch); RCCL is used only to scatter the problem data from rank 0 and to all-gather u*.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NX, NU, NP = 12, 4, 30
XBOX = 10.0
HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md chip-level parameters (achievable: ~6.3e12)
U_ERR_SAMPLE = 32          # instances whose u* is compared with the tight-tolerance CPU reference


def make_instances(first, count):
    from pympc_amd import fixtures
    kws = [fixtures.random_lti(first + i, nx=NX, nu=NU, Np=NP, xbox=XBOX) for i in range(count)]
    return {k: np.stack([np.asarray(kw[k], dtype=float) for kw in kws]) for k in ('Ad', 'Bd', 'x0')}


def algorithmic_bytes_8d(n, m, nnzL, iters, checks, solves):
    """a generic sparse-LDL' ADMM, FP64 values only.  per iteration 8(2 nnzL_strict + 6n + 10m); per
    residual evaluation 8(nnz(triu P) + 2 nnz(A)); per solve 8(4n + 6m).  Kept beside the design figure: it charges the
    iterate and metric vectors to HBM every iteration, which this implementation keeps in LDS/registers."""
    nnz_triuP = (NP + 1) * NX * 2 + NP * NU + (NP - 1) * NU     # diagonal weights: diag + upper QDu coupling
    nnzA = (NP + 1) * NX + NP * NX * NX + NP * NX * NU + 2 * (NP + 1) * NX + NP * NU + NU + 2 * NP * NU - 1
    b_it = 8 * (2 * (nnzL - n) + 6 * n + 10 * m)
    return iters * b_it + checks * 8 * (nnz_triuP + 2 * nnzA) + solves * 8 * (4 * n + 6 * m), b_it


def pmc_bytes_per_iter(workload, path, kernel):
    """Measured memory-side bytes per ADMM iteration per instance of `kernel`, from the committed rocprofv3 PMC passes of
    this same command (profiles/pmc_hbm_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate runs, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950, divided by the ADMM iterations of the profiled launches).
    Counters cannot be read from inside the process; scaling the per-iteration figure by this run's iteration count
    gives the per-launch traffic for whatever launch length the caller chose."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_hbm_traffic.json')) as f:
            return json.load(f)[workload][path][kernel]['hbm_bytes_per_iter_per_qp']
    except Exception:
        return None


def cpu_legs(eps, samples, want_baseline, seconds_budget=12.0, inst=400, steps=100):
    """Everything that needs the CPU oracle (test infrastructure; the only place bench.py touches oracle/):
    (1) cpu_baseline -- the reference-style CPU path on this box's host cores: the C port of the OSQP algorithm
        (oracle/osqp_ref.c, rebuilt here with -O3 -march=native) driven by its C closed-loop driver like the reference
        drives OSQP: 1 thread, sequential over instances, warm-started receding horizon on the same workload recipe
        (`value`, what pyMPC does today), and the same on every usable core at once;
    (2) u*_ref of the sampled QPs at tolerance 1e-10, for 'max |u* - u*_ref|'."""
    from oracle import cpu_bench
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100, help='timed MPC steps (SURVEY 8d cfg-3: 100-step receding horizon)')
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=None, help='instances per GPU (weak scaling; default 1024, cfg5: 512)')
    ap.add_argument('--total-batch', type=int, default=None, help='instances in total, split evenly over the GPUs (strong scaling)')
    ap.add_argument('--eps', type=float, default=1e-3)
    ap.add_argument('--chunk', type=int, default=None, help='device loop: steps per kernel launch (default: the timed steps in equal launches of at most 25)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-path', action='store_true', help='skip the secondary measurements (other path, parity setting)')
    ap.add_argument('--no-refactor-timing', action='store_true', help='skip the timing of the factorization alone (profiling runs: its launches carry the solve kernel\'s name)')
    ap.add_argument('--path', default='device_loop', choices=['stepwise', 'device_loop'],
                    help='stepwise: update()/solve()/output() per step from the host (the reference call pattern); '
                         'device_loop: the same K steps inside mpcqp_mpc_loop (SURVEY 8f-1)')
    ap.add_argument('--workload', default='cfg3', choices=['cfg3', 'cfg5'],
                    help='cfg3: 1024 x (12,4,30) (headline); cfg5: 512 x (20,8,100), tight state box (SURVEY 8d)')
    args = ap.parse_args()
    global NX, NU, NP, XBOX
    if args.workload == 'cfg5':
        NX, NU, NP, XBOX = 20, 8, 100, 1.0

    import torch
    import torch.distributed as dist
    from pympc_amd.solver import BatchProblem

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an AMD GPU; pympc_amd has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', device_id=dev)
    if args.total_batch is not None:
        if args.total_batch % world:
            raise SystemExit('--total-batch must be divisible by the number of GPUs')
        B, scaling = args.total_batch // world, 'strong'
    else:
        B, scaling = (args.batch if args.batch is not None else (1024 if args.workload == 'cfg3' else 512)), 'weak'
    f64 = torch.float64

    # ---- problem data: generated on rank 0, scattered over RCCL (north_star: scatter problem data)
    from pympc_amd import sharding
    full = None
    if rank == 0:
        full = {k: torch.from_numpy(v).to(dev) for k, v in make_instances(0, B * world).items()}
    loc = sharding.scatter_instances(full, {'Ad': (NX, NX), 'Bd': (NX, NU), 'x0': (NX,)}, B, dev)
    Ad, Bd, x = loc['Ad'], loc['Bd'], loc['x0'].clone()

    stream = torch.cuda.current_stream(dev)
    prob = BatchProblem(B, NX, NU, NP, device=local_rank, stream=stream.cuda_stream,
                        eps_abs=args.eps, eps_rel=args.eps, warm_start=1)
    eye = lambda k, s: (s * torch.eye(k, dtype=f64, device=dev)).expand(B, k, k).contiguous()
    ones = lambda k, s: torch.full((B, k), s, dtype=f64, device=dev)
    prob.setup(Ad, Bd, eye(NX, 1.0), eye(NX, 1.0), eye(NU, 0.1), eye(NU, 0.1),
               ones(NX, -XBOX), ones(NX, XBOX), ones(NU, -1.0), ones(NU, 1.0), ones(NU, -0.5), ones(NU, 0.5),
               ones(NU, 0.0), torch.full((B, 1), 1e6, dtype=f64, device=dev),
               x, ones(NU, 0.0), torch.zeros((B, NX), dtype=f64, device=dev))
    prob.solve_async()                        # cold solve (setup(solve=True))
    u = torch.empty((B, NU), dtype=f64, device=dev)
    prob.u0(out=u)
    # whole-process work per kernel (for the profile scripts: counter totals / these = bytes per iteration): 'solve' =
    # k_mpc_run<..,false> (mpcqp_solve), 'loop' = k_mpc_run<..,true> (mpcqp_mpc_loop)
    totals = {'solve': [0, 0, 0], 'loop': [0, 0, 0]}

    def account(kind, st=None):
        st = prob.stats(reset=True) if st is None else st
        for i, v in enumerate((st[0], st[1], st[3])):
            totals[kind][i] += v
        return st
    account('solve')
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    u_all = torch.empty((world * B, NU), dtype=f64, device=dev) if world > 1 else None

    def plant(xc, uc):
        w = 0.01 * torch.randn((B, NX), dtype=f64, device=dev, generator=gen)
        return torch.baddbmm(w.unsqueeze(2), Ad, xc.unsqueeze(2)).add_(torch.bmm(Bd, uc.unsqueeze(2))).squeeze(2)

    def timed(kind, run_warm, run_timed):
        """W untimed steps, then the timed K steps between barrier + synchronize; returns max-over-ranks seconds
        and the device-side accounting of the timed region."""
        run_warm()
        account(kind)
        prob.profile(enable=True, reset=True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_timed()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=f64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        iters, checks, refacts, solves = account(kind)
        run_ms, launches = prob.profile(enable=False)
        return dict(elapsed=elapsed, iters=iters, checks=checks, refacts=refacts, solves=solves, run_ms=run_ms, launches=launches)

    def measure_stepwise(steps, warmup):
        """The reference's call pattern: per step the host calls update(), solve(), output() (one kernel launch per
        solve); plant and disturbance are torch ops on the same stream; with N > 1 u* is all-gathered every step."""
        nonlocal x

        def step():
            nonlocal x
            x = plant(x, u)
            prob.update(x, u)
            prob.solve_async()
            prob.u0(out=u)
            if world > 1:
                sharding.gather_inputs(u, out=u_all)

        return timed('solve', lambda: [step() for _ in range(warmup)], lambda: [step() for _ in range(steps)])

    def measure_device_loop(steps, warmup):
        """The same closed loop inside mpcqp_mpc_loop: launches of `chunk` steps each, so that every
        launch (warm-up and timed) does the same work; the disturbance sequence is synthetic input generated before
        the timed region; with N > 1 the applied inputs of a chunk are all-gathered after it."""
        nonlocal x
        # steps per launch: a launch ends when its slowest instance has finished its steps (no instance can run ahead of its
        # own closed loop), so short launches pay the spread of the per-instance iteration counts more often -- at cfg-3,
        # 5-step launches cost 8 % against 20-step ones.  Default: the whole timed region in launches of at most 25 steps.
        if args.chunk:
            chunk = args.chunk
        else:
            chunk = steps
            while chunk > 25:
                chunk = next((chunk // d for d in (2, 3, 5, 7) if chunk % d == 0), 25)
        w_all = 0.01 * torch.randn((warmup + steps, B, NX), dtype=f64, device=dev, generator=gen)
        outs = (torch.empty((chunk + 1, B, NX), dtype=f64, device=dev), torch.empty((chunk, B, NU), dtype=f64, device=dev),
                torch.empty((chunk, B), dtype=torch.int32, device=dev), torch.empty((chunk, B), dtype=torch.int32, device=dev))
        u_hist = torch.empty((world * chunk, B, NU), dtype=f64, device=dev) if world > 1 else None

        def run(first, count):
            for c in range(first, first + count, chunk):
                k = min(chunk, first + count - c)
                o = outs if k == chunk else tuple(t[:k + (1 if i == 0 else 0)] for i, t in enumerate(outs))
                prob.mpc_run(k, w=w_all[c:c + k], out=o)
                if world > 1 and k == chunk:
                    sharding.gather_trajectory(outs[1], out=u_hist)
            return o

        last = {}
        r = timed('loop', lambda: last.update(o=run(0, warmup)) if warmup else None, lambda: last.update(o=run(warmup, steps)))
        x = last['o'][0][-1].clone()
        u.copy_(last['o'][1][-1])
        r['chunk'] = chunk
        return r

    def sample_point():
        """One more (untimed) closed-loop step through the stepwise API: returns the sampled QPs (x0, u_{-1}) and the
        u* the device produced for them at the current tolerance.  Rank 0's instances only (global index = local)."""
        nonlocal x
        x = plant(x, u)
        um1 = u.clone()
        prob.update(x, um1)
        prob.solve_async()
        prob.u0(out=u)
        torch.cuda.synchronize()
        account('solve')
        idx = np.unique(np.linspace(0, B - 1, min(U_ERR_SAMPLE, B)).astype(int))
        infos = prob.infos()
        return dict(idx=idx, x0=x[idx].cpu().numpy(), um1=um1[idx].cpu().numpy(), u=u[idx].cpu().numpy(),
                    solved=np.array([infos[int(i)].status == 1 for i in idx]))

    measure = {'stepwise': measure_stepwise, 'device_loop': measure_device_loop}
    res = measure[args.path](args.steps, args.warmup)
    elapsed, iters, checks, refacts, solves = res['elapsed'], res['iters'], res['checks'], res['refacts'], res['solves']
    admm_ms, admm_launches = res['run_ms'], res['launches']
    infos = prob.infos()
    n_solved = sum(1 for i in infos if i.status == 1)
    samples = [dict(sample_point(), eps=args.eps)]
    other = None
    if not args.no_other_path:
        oname = 'stepwise' if args.path == 'device_loop' else 'device_loop'
        o = measure[oname](args.steps, args.warmup)
        other = {'path': oname, 'value': B * world * args.steps / o['elapsed'], 'ms_per_step': 1e3 * o['elapsed'] / args.steps,
                 'mean_admm_iters': o['iters'] / max(1, o['solves'])}
    parity = None
    if not args.no_other_path and args.eps > 1e-8:
        # the same loop at the parity setting eps = 1e-9 (the tolerance the u* comparison is made at)
        prob.update_settings(eps_abs=1e-9, eps_rel=1e-9)
        pr = measure[args.path](args.steps, args.warmup)
        pinf = prob.infos()
        parity = {'eps_abs': 1e-9, 'eps_rel': 1e-9, 'path': args.path, 'value': B * world * args.steps / pr['elapsed'],
                  'ms_per_step': 1e3 * pr['elapsed'] / args.steps, 'mean_admm_iters': pr['iters'] / max(1, pr['solves']),
                  'solved_fraction_last_step': sum(1 for i in pinf if i.status == 1) / B}
        samples.append(dict(sample_point(), eps=1e-9))
        prob.update_settings(eps_abs=args.eps, eps_rel=args.eps)
    # what one rho update costs: the block factorization of every instance, timed alone (mpcqp_refactor rewrites the factor
    # that is already in place); the steady-state loop above needs none, the cold solve a few per instance
    refactor_ms = None
    if not args.no_refactor_timing:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        prob.refactor(); torch.cuda.synchronize()
        ev0.record()
        for _ in range(3):
            prob.refactor()
        ev1.record(); torch.cuda.synchronize()
        refactor_ms = ev0.elapsed_time(ev1) / 3
    kname = prob.kernel_name(loop=args.path == 'device_loop')
    lds_state = ',true,' in kname.split('<')[1][:9]            # second template argument: iterate resident in LDS

    if rank == 0:
        n, m, nnzL = prob.n, prob.m, prob.nnzL
        per_iter, per_round, per_solve = prob.stream_bytes()
        # the one kernel of the path, k_mpc_run, does everything (QP refresh, ADMM iterations, residual checks).  HIP events
        # bracket every launch on its stream (mpcqp_profile); the bytes are what this implementation streams by design.
        design_bytes = iters * per_iter + checks * per_round + solves * per_solve
        achieved = design_bytes / (admm_ms * 1e-3)
        alg8d, b_it8d = algorithmic_bytes_8d(n, m, nnzL, iters, checks, solves)
        pmc = pmc_bytes_per_iter(args.workload, args.path, kname)
        traffic = pmc * iters / max(1, admm_launches) if pmc else None
        out = {
            'metric': 'QP-solves/sec (MPC steps/sec) at nx=%d nu=%d Np=%d; max |u*-u*_ref|' % (NX, NU, NP),
            'value': B * world * args.steps / elapsed,
            'unit': 'QP-solves/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%s: %d random stable LTI MPC instances per GPU (nx=%d, nu=%d, Np=Nc=%d, n=%d, m=%d), '
                                   'warm-started receding horizon x+=Ad x+Bd u*+w' % ('cfg-3' if args.workload == 'cfg3' else 'cfg-5', B, NX, NU, NP, n, m),
                       'batch_per_gpu': B, 'total_batch': B * world, 'eps_abs': args.eps, 'eps_rel': args.eps, 'path': args.path,
                       'parallelism': 'instances sharded over %d GPU(s); RCCL scatter of data, all-gather of u*' % world},
            'mean_admm_iters': iters / max(1, solves),
            'solved_fraction_last_step': n_solved / B,
            'refactorizations_per_solve': refacts / max(1, solves),
            'refactorization': {'per_solve_timed_region': refacts / max(1, solves), 'ms_per_batch': refactor_ms, 'us_per_instance_amortised': (1e3 * refactor_ms / B) if refactor_ms else None,
                                'note': 'block LDL factorization of all %d instances in one launch (one rho update each); 0 per solve in the warm '
                                        'receding-horizon loop, a few per instance during the cold solve' % B},
            'roofline': {'bound': 'hbm', 'achieved': achieved / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK, 'traffic': traffic,
                         'traffic_GBps': (traffic / (admm_ms / max(1, admm_launches) * 1e-3) / 1e9) if traffic else None,
                         'bytes_model': 'design: what k_mpc_run streams per instance (mpcqp_get_stream_bytes) -- per ADMM iteration the KKT factor '
                                        '(%s%s), per round the residual-evaluation inputs and the '
                                        'iterate in/out of LDS, per solve the QP refresh and the write-out'
                                        % ('forward matrices of N-1 stages, packed S^-1 of N stages and the stage tables once each, [G|G\'] once' if NX + NU <= 16
                                           else 'packed S^-1 of N stages twice, one stage table per sweep, [G|G\'] by each sweeping wave',
                                           '' if lds_state else '; iterate and metric vectors too: they do not fit LDS at this size'),
                         'design_bytes_per_iter_per_qp': per_iter, 'design_bytes_per_round_per_qp': per_round, 'design_bytes_per_solve_per_qp': per_solve,
                         'design_bytes_per_launch': design_bytes / max(1, admm_launches),
                         'measured_bytes_per_iter_per_qp': pmc,
                         'kernel': kname, 'kernel_ms': admm_ms / max(1, admm_launches), 'launches': admm_launches,
                         'steps_per_launch': res.get('chunk', 1),
                         'algorithmic_8d': {'bytes_per_launch': alg8d / max(1, admm_launches), 'bytes_per_iter_per_qp': b_it8d, 'nnzL': nnzL,
                                            'GBps': alg8d / (admm_ms * 1e-3) / 1e9,
                                            'note': 'SURVEY 8(d) generic sparse-LDL formula; charges 6n+10m vector doubles per iteration to HBM that this '
                                                    'kernel keeps in LDS/registers, so it may exceed the HBM peak -- not used for frac'}},
            'accounting': {'timed': {'iters': iters, 'rounds': checks, 'solves': solves, 'launches': admm_launches, 'kernel_ms_total': admm_ms},
                           'process_totals': {prob.kernel_name(loop=(k == 'loop')): dict(iters=v[0], rounds=v[1], solves=v[2]) for k, v in totals.items()}},
            'other_path': other,
            'parity_setting': parity,
        }
        cpu, refs = cpu_legs(args.eps, samples if world == 1 else [], want_baseline=(not args.no_cpu_baseline and world == 1))
        if cpu:
            out['cpu_baseline'] = cpu
        if refs:
            err = {}
            for s, ref in zip(samples, refs):
                ok = s['solved'] & np.isfinite(ref).all(axis=1)
                d = np.abs(s['u'][ok] - ref[ok]).max() if ok.any() else float('nan')
                sc = max(1e-3, np.abs(ref[ok]).max()) if ok.any() else 1.0
                err['eps_%g' % s['eps']] = {'max_abs': float(d), 'max_rel': float(d / sc), 'instances': int(ok.sum())}
            out['u_err'] = dict(err, definition='max over the sample of |u* - u*_ref|_inf; rel = / max |u*_ref|_inf',
                                reference='oracle/osqp_ref.c at eps 1e-10 on the same (x0, u_-1): the QP the device solved in one more warm-started step',
                                north_star_tolerance_rel=1e-6)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
