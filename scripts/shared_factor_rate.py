"""One model, many states (test_scripts/example_mpc_function.py:105-111) as a closed loop: B instances of ONE random (12,4,30) model, set up from one common
state (what one reference controller's setup() is), then scattered states and per-instance noise.  Measures the device loop with every instance streaming its
own factor against mpcqp_share_factor (one shared copy: L2 instead of HBM), and checks that the two give bit-identical trajectories.
    python scripts/shared_factor_rate.py [--batch 1024] [--steps 40] [--backend sweeps]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(args, B, backend, share, torch, dev):
    from pympc_amd.solver import BatchProblem
    from pympc_amd import fixtures, _lib
    NX, NU, NP, XBOX = args.nx, args.nu, args.np, args.xbox
    f64 = torch.float64
    kw = fixtures.random_lti(args.seed, nx=NX, nu=NU, Np=NP, xbox=XBOX)
    Ad, Bd, x_common = (np.asarray(kw[k], dtype=float) for k in ('Ad', 'Bd', 'x0'))
    rng = np.random.default_rng(args.seed + 1)
    X0 = x_common[None, :] * rng.uniform(0.2, 1.0, size=(B, 1)) * rng.choice([-1.0, 1.0], size=(B, NX))      # scattered states
    W = torch.from_numpy(0.01 * rng.standard_normal((args.warmup + args.steps, B, NX))).to(dev)
    stream = torch.cuda.current_stream(dev)
    prob = BatchProblem(B, NX, NU, NP, device=dev.index, stream=stream.cuda_stream, eps_abs=args.eps, eps_rel=args.eps, warm_start=1, backend=backend,
                        tuning=0 if share else _lib.TUNE_NO_SHARE)      # (setup shares by itself unless told not to)
    ones = lambda k, s: np.full((B, k), s)
    prob.setup(Ad, Bd, np.eye(NX), np.eye(NX), 0.1 * np.eye(NU), 0.1 * np.eye(NU), ones(NX, -XBOX), ones(NX, XBOX), ones(NU, -1.0), ones(NU, 1.0),
               ones(NU, -0.5), ones(NU, 0.5), ones(NU, 0.0), np.full((B, 1), 1e6), np.broadcast_to(x_common, (B, NX)), ones(NU, 0.0), np.zeros((B, NX)))
    prob.solve_async(); prob.synchronize()
    nshared = prob.share_factor() if share else 0
    prob.update(X0, np.zeros((B, NU)))
    prob.solve_async(); prob.synchronize()
    prob.stats(reset=True)
    chunk = args.chunk
    outs = (torch.empty((chunk + 1, B, NX), dtype=f64, device=dev), torch.empty((chunk, B, NU), dtype=f64, device=dev),
            torch.empty((chunk, B), dtype=torch.int32, device=dev), torch.empty((chunk, B), dtype=torch.int32, device=dev))
    hist = []
    for c in range(0, args.warmup, chunk):
        prob.mpc_run(chunk, w=W[c:c + chunk], out=outs); hist.append(outs[1].clone())
    torch.cuda.synchronize()
    prob.stats(reset=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for c in range(args.warmup, args.warmup + args.steps, chunk):
        prob.mpc_run(chunk, w=W[c:c + chunk], out=outs); hist.append(outs[1].clone())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    st = prob.stats(reset=True)
    U = torch.cat(hist).cpu().numpy()
    res = dict(backend=backend, share=bool(share), nshared=nshared, batch=B, solves_per_s=B * args.steps / (ms * 1e-3), ms=ms, iters_per_solve=st[0] / max(1, st[3]),
               refactorizations=st[2], kernel=prob.kernel_name(True), solved=float((outs[2] == 1).double().mean()))
    prob.close()
    return res, U


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, nargs='+', default=[1024])
    ap.add_argument('--steps', type=int, default=40); ap.add_argument('--warmup', type=int, default=20); ap.add_argument('--chunk', type=int, default=20)
    ap.add_argument('--nx', type=int, default=12); ap.add_argument('--nu', type=int, default=4); ap.add_argument('--np', type=int, default=30)
    ap.add_argument('--xbox', type=float, default=None); ap.add_argument('--eps', type=float, default=1e-3); ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--backend', nargs='+', default=['sweeps'])
    args = ap.parse_args()
    import torch
    if args.xbox is None:
        args.xbox = 10.0
    dev = torch.device('cuda', 0)
    for B in args.batch:
        for backend in args.backend:
            r0, U0 = run(args, B, backend, False, torch, dev)
            r1, U1 = run(args, B, backend, True, torch, dev)
            same = bool(np.array_equal(U0, U1))
            print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r0.items()})
            print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r1.items()}, 'bit_identical', same, 'speedup %.3f' % (r1['solves_per_s'] / r0['solves_per_s']), flush=True)
        ra, _ = run(args, B, 'auto', False, torch, dev)
        print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in ra.items()}, flush=True)


if __name__ == '__main__':
    main()
