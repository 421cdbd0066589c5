// mpcqp_kernels.h -- part of libmpcqp_hip (included by mpcqp.hip, one translation unit).
// The persistent solve / closed-loop kernel k_mpc_run and the verification kernels.
#pragma once

// ------------------------------------------------------------------------------------------------
// Device-side receding-horizon loop (the caller pattern of examples/example_point_mass.py:88-101 and
// pyMPC/mpc.py:688-692):   for k in range(K):  u = K.output();  x = Ap x + Bp u + w_k;  K.update(x)
// One workgroup walks its own instance through all K steps -- output (mpc.py:271-336, u_failure = uref unless
// 'solved'), plant, QP refresh (mpc.py:386-454), warm-started solve -- with no host round trip and, unlike the
// per-step API, no batch-wide barrier per round: an instance that needs 50 iterations does not hold up one that
// needs 25.  The same kernel with nsteps = 0 is
// mpcqp_solve: begin, rounds of { admm, check } until this instance terminates -- no host loop, no batch barrier.
// ------------------------------------------------------------------------------------------------
// (RunArgs, RunKArgs, run_kargs(), next_stop / stop_mode: mpcqp_run.h -- the latency round reads them too)

struct RunSmem { Smem S; double *X, *Z, *Y; };
template <bool LDSSTATE>
__device__ __forceinline__ RunSmem run_smem(const Lay &L, const Ptrs &P) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    RunSmem r; double *p = sh; smem_common(L, P, p, r.S);
    r.X = r.Z = r.Y = nullptr;
    if (LDSSTATE) { r.X = carve(p, L.n); r.Z = carve(p, L.m); r.Y = carve(p, L.m); }
    return r;
}

// (OCC: workgroups per CU of the calling kernel -- a tag only: it keeps the phases of kernels with different launch bounds apart,
//  so that the register budget of the one-workgroup-per-CU kernels does not leak into the four-per-CU ones through a shared callee)
// (the grouped small stages' factorization: a function of its own, see factor_grouped in mpcqp_group.h)
template <int OCC>
__device__ __noinline__ void run_group_factor_phase(int *frame_pin) {
    PHASE_PIN_USE(frame_pin);
    const RunKArgs &A = run_kargs();
    const Lay &L = A.L; const Ptrs &P = A.P;
    RunSmem r = run_smem<false>(L, P);
    const int b = inst_of(P.perm);
    factor_unshare(P, b);
    Ctx c{L, r.S.hot, P.model + (size_t)b * L.model_sz + L.hot_sz};
    factor_grouped(c, P.omega + (size_t)b * L.m, P.s + (size_t)b * L.n, P.c[b], P.F + (size_t)b * P.fsz, r.S.T, r.S.iflag, border_ptrs(L, P, r.S));
}

template <int NB, int OCC>
__device__ __noinline__ void run_factor_phase(int *frame_pin) {
    PHASE_PIN_USE(frame_pin);
    const RunKArgs &A = run_kargs();
    const Lay &L = A.L; const Ptrs &P = A.P;
    if constexpr (NB == 16 && !kLatOnly) { if (L.grp > 1) { FramePin pin; run_group_factor_phase<OCC>(&pin.v); return; } }
    RunSmem r = run_smem<false>(L, P);                       // (the common LDS area comes first in both layouts)
    const int b = inst_of(P.perm);
    factor_unshare(P, b);
    Ctx c{L, r.S.hot, P.model + (size_t)b * L.model_sz + L.hot_sz};
    if constexpr (kLatOnly) {                                // (mpcqp_w8.hip: cyclic-reduction handles, or grouped small stages)
        if (L.bcr) factor_bcr(c, P.omega + (size_t)b * L.m, P.s + (size_t)b * L.n, P.c[b], P.F + (size_t)b * P.fsz, P.bws + (size_t)b * L.bcr * BcrFmt::WSTAGE, r.S.T, r.S.iflag);
        else factor_grouped(c, P.omega + (size_t)b * L.m, P.s + (size_t)b * L.n, P.c[b], P.F + (size_t)b * P.fsz, r.S.T, r.S.iflag, border_ptrs(L, P, r.S));
        return;
    }
    if (NB == 16 && L.dense) factor_dense(c, P.omega + (size_t)b * L.m, P.s + (size_t)b * L.n, P.c[b], P.F + (size_t)b * P.fsz, r.S.T, r.S.iflag);      // (the register-resident backends: 16 x 16 stages only)
    else if (NB == 16 && L.bcr) factor_bcr(c, P.omega + (size_t)b * L.m, P.s + (size_t)b * L.n, P.c[b], P.F + (size_t)b * P.fsz, P.bws + (size_t)b * L.bcr * BcrFmt::WSTAGE, r.S.T, r.S.iflag);
    else factor_all<NB>(c, P.omega + (size_t)b * L.m, P.s + (size_t)b * L.n, P.c[b], P.F + (size_t)b * P.fsz, r.S.T, r.S.iflag,
                        border_ptrs(L, P, r.S));
}

// iter0: iterations of this solve done so far.  Returns 1 if the phase has FINISHED the solve itself (the latency round with a dense top runs OSQP's
// termination test in its own layout, round after round: admm_latw) -- the iteration count reached is then in Smem::iflag[5], as it is whenever
// that round returns -- and 0 if the generic check is to follow.
template <int NB, bool LDSSTATE, int NXT, int NUT, int MODE>
__device__ __forceinline__ int run_admm_phase_body(int iters, int iter0) {
    const RunKArgs &A = run_kargs();
    const Lay &L = A.L; const Ptrs &P = A.P;
    RunSmem r = run_smem<LDSSTATE>(L, P);
    HotPtrs hp; hp.model = P.model; hp.step = P.step; hp.omega = P.omega; hp.s = P.s; hp.qv = P.qv; hp.F = P.F; hp.c = P.c;
    hp.Bb = P.Bb; hp.Zb = P.Zb; hp.Sig = P.Sig; hp.x = P.x; hp.z = P.z; hp.y = P.y; hp.dx = P.dx; hp.dy = P.dy; hp.perm = P.perm; hp.fsz = P.fsz; hp.fown = P.fown;
    if constexpr (MODE >= MODE_BCRT)
        return admm_latw<NXT, NUT, MODE - MODE_BCRT>(L, hp, r.S, r.X, r.Z, r.Y, A.S.alpha, __builtin_amdgcn_readfirstlane(iters), __builtin_amdgcn_readfirstlane(iter0));
    admm_body<NB, LDSSTATE, NXT, NUT, MODE>(L, hp, r.S, r.X, r.Z, r.Y, A.S.alpha, __builtin_amdgcn_readfirstlane(iters));
    return 0;
}
// (frame_pin: the address of a local of the CALLER, made opaque here.  A call that may reach into its caller's frame cannot be marked `tail`,
//  and for an internal, non-recursive function none of whose calls is a tail call LLVM's interprocedural register allocation treats NO register
//  as callee-saved (TargetFrameLowering::isSafeForNoCSROpt): the phase then starts without saving the ~ 340 (latency kernels) / 48 (bandwidth
//  kernels) registers of the calling convention's callee-saved set -- the kernel keeps the handful of values it has live across the call itself.)
template <int NB, bool LDSSTATE, int NXT, int NUT, int MODE>
__device__ __noinline__ int run_admm_phase(int iters, int iter0, int *frame_pin) {
    PHASE_PIN_USE(frame_pin);
    return run_admm_phase_body<NB, LDSSTATE, NXT, NUT, MODE>(iters, iter0);
}
// (History: as an ordinary call the latency kernels' phase saved and restored ~ 340 callee-saved registers per call -- 46 KB per iteration and
//  instance in the write counters of a kernel that by design reads nothing; taking the phase INLINE removed that traffic but made the allocator
//  spill inside the iteration loop, 914 k -> 885 k solves/s at 256 instances.  With frame_pin the call stays and the saves are gone: 941 k.)
template <int NB, bool LDSSTATE, int NXT, int NUT, int MODE, int OCC>
__device__ __forceinline__ int run_admm(int iters, int iter0) { FramePin pin; return run_admm_phase<NB, LDSSTATE, NXT, NUT, MODE>(iters, iter0, &pin.v); }

template <int NB, bool LDSSTATE, int OCC>
__device__ __noinline__ void run_begin_phase(int plain, int warm_x, int *frame_pin) {
    PHASE_PIN_USE(frame_pin);
    const RunKArgs &A = run_kargs();
    RunSmem r = run_smem<LDSSTATE>(A.L, A.P);
    begin_body<NB, OCC>(A.L, A.P, A.S, r.S, __builtin_amdgcn_readfirstlane(plain), __builtin_amdgcn_readfirstlane(warm_x));
}

template <int NB, bool LDSSTATE, int OCC>
__device__ __noinline__ int run_check_phase(int iter, int mode, int *frame_pin) {
    PHASE_PIN_USE(frame_pin);
    const RunKArgs &A = run_kargs();
    RunSmem r = run_smem<LDSSTATE>(A.L, A.P);
    return check_body<NB, OCC>(A.L, A.P, A.S, r.S, __builtin_amdgcn_readfirstlane(iter), __builtin_amdgcn_readfirstlane(mode), r.X, r.Z, r.Y);
}

constexpr int run_occupancy(int NB, int MODE) { return (NT > 256 || MODE == MODE_DENSE || MODE >= MODE_BCR || NB > 32) ? 1 : NB <= 16 ? 4 : 2; }      // workgroups per CU (512-thread kernels: one, two waves per SIMD)
// One instance from the kernel's prologue to its last store: what a workgroup of k_mpc_run does ONCE when the grid is the batch, and once per instance it
// takes off the queue when the launch is persistent (below).
// LOOP: the closed-loop steps [k0, k1) of the launch's nsteps (the whole launch, or the part of it one queue item covers: the state an instance carries
// from step to step is in memory at every step boundary -- step blob, iterate, mpcqp_info -- exactly as between two launches).
template <int NB, bool LDSSTATE, int NXT, int NUT, int MODE, bool LOOP>
__device__ __forceinline__ void run_instance(const int k0, const int k1) {
    constexpr int OCC = run_occupancy(NB, MODE);
    const RunKArgs &A = run_kargs();
    const Lay &L = A.L; const Ptrs &P = A.P; const RunArgs &R = A.R;
    RunSmem rs = run_smem<LDSSTATE>(L, P);
    Smem &S = rs.S;
    const int b = inst_of(P.perm), tid = threadIdx.x;
    double *step = P.step + (size_t)b * L.step_sz;
    if (LOOP && tid == 0 && k0 == 0) P.tstamp[(size_t)TS_STRIDE * b] = wall_clock64();
    if (LOOP && tid == 0 && k0 == 0 && R.nsteps < TS_STEPS) {      // (WHERE the instance's first steps ran -- XCC and HW_ID in the last step's slot of its stamps; scripts/diag_makespan.py)
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        P.tstamp[(size_t)TS_STRIDE * b + 2 + TS_STEPS - 1] = ((unsigned long long)(xcc & 15) << 32) | hw;
    }
    if (MODE >= MODE_BCRT && tid == 0) S.iflag[2] = 0;      // (the rounds' LDS-resident part of the factor is not loaded yet: admm_latw; a barrier follows in load_common)
    const double *step_src = step;
    if (!LOOP && R.inl_n) {                      // update(x0, u_{-1}, xref) from the kernel arguments: into the step blob for the phases that follow (behind load_common's
        for (int i = tid; i < R.inl_n; i += NT) step[i] = R.inl[i];      // barrier), while load_common reads the very same values from the arguments -- no round trip at all
        step_src = R.inl;
    } else if (!LOOP && R.pin_in) {              // ... or straight from the caller's (mapped) memory: one PCIe round trip
        const double *src = R.pin_in + (size_t)b * R.pin_stride;
        if (R.pin_mask & 1) for (int i = tid; i < L.nx; i += NT) step[i] = src[i];
        if (R.pin_mask & 2) for (int i = tid; i < L.nu; i += NT) step[L.nx + i] = src[L.nx + i];
        if (R.pin_mask & 4) for (int i = tid; i < R.pin_xref; i += NT) step[L.nx + L.nu + i] = src[L.nx + L.nu + i];
        __syncthreads();
    }
    load_common(L, P.model + (size_t)b * L.model_sz, step_src, S);
    if (!LOOP && R.part == 3) {                  // mpcqp_refactor: the factorization alone (what one rho update costs)
        __syncthreads();
        { FramePin pin; run_factor_phase<NB, OCC>(&pin.v); }
        return;
    }
    const int nx = L.nx, nu = L.nu;
    for (int k = k0; k < k1; ++k) {              // (LOOP = false: [0, 1), one solve of the current data -- mpcqp_solve)
        if (LOOP) {
            // scratch in the (idle) work area: un | xn | xt | ym | inn | xu, 128 doubles each (nx + nu <= 128, ny <= 64)
            double *un = S.T, *xn = S.T + 128, *xt = S.T + 256, *ym = S.T + 384, *inn = S.T + 512, *xu = S.T + 640;
            const size_t kb = (size_t)k * R.batch + b;
            const int ny = R.ny;
            // ---- output(): first input of the current solution, or u_failure
            // (from the second step on, status and first input of the solve that just ended are where its last check left them in LDS)
            const int status = k > k0 ? S.iflag[4] : P.info[b].status;
            if (tid < nu) un[tid] = status == MPCQP_SOLVED ? (k > k0 ? S.uo[tid] : P.xo[(size_t)b * L.n + L.ou + tid]) : S.hot[L.ouref + tid];
            if (tid < nx) xt[tid] = ny ? R.x_true[(size_t)b * nx + tid] : S.x0s[tid];      // the plant state
            __syncthreads();
            if (ny && tid < ny) {                            // measurement y = C x + v and innovation y - C xhat
                const double *C = R.C + (size_t)b * ny * nx + (size_t)tid * nx;
                double y = R.v ? R.v[kb * ny + tid] : 0.0, yh = 0.0;
                for (int j = 0; j < nx; ++j) { y += C[j] * xt[j]; yh += C[j] * S.x0s[j]; }
                ym[tid] = y; inn[tid] = y - yh;
                if (R.y_traj) R.y_traj[kb * ny + tid] = y;
            }
            // ---- plant step
            if (tid < nx) {
                const double *Ap = R.Ap ? R.Ap + (size_t)b * nx * nx : S.hot + L.oAd;
                const double *Bp = R.Bp ? R.Bp + (size_t)b * nx * nu : S.hot + L.oBd;
                double v = R.w ? R.w[kb * nx + tid] : 0.0;
                double acc = 0.0;
                for (int j = 0; j < nx; ++j) acc += Ap[tid * nx + j] * xt[j];
                for (int j = 0; j < nu; ++j) acc += Bp[tid * nu + j] * un[j];
                xn[tid] = acc + v;
                R.x_traj[kb * nx + tid] = xt[tid];
                if (ny) { R.x_true[(size_t)b * nx + tid] = xn[tid]; if (R.xhat_traj) R.xhat_traj[kb * nx + tid] = S.x0s[tid]; }
            }
            if (tid < nu) R.u_traj[kb * nu + tid] = un[tid];
            __syncthreads();
            if (ny) {                                        // KF.update(y): xhat[k|k] = xhat[k|k-1] + L (y - yhat);  KF.predict(u)
                if (tid < nx) {
                    const double *Lg = R.Lg + (size_t)b * nx * ny + (size_t)tid * ny;
                    double acc = S.x0s[tid];
                    for (int j = 0; j < ny; ++j) acc += Lg[j] * inn[j];
                    xu[tid] = acc;
                }
                __syncthreads();
                if (tid < nx) {
                    const double *Ad = S.hot + L.oAd, *Bd = S.hot + L.oBd;
                    double acc = 0.0;
                    for (int j = 0; j < nx; ++j) acc += Ad[tid * nx + j] * xu[j];
                    for (int j = 0; j < nu; ++j) acc += Bd[tid * nu + j] * un[j];
                    xn[tid] = acc;                           // xhat[k+1|k]: what the controller is updated with
                }
                __syncthreads();
            }
            // ---- update(x): new initial state, previous input (mpc.py:338-364) and, if given, reference
            if (tid < nx) { S.x0s[tid] = xn[tid]; step[tid] = xn[tid]; }
            if (tid < nu) { S.um1s[tid] = un[tid]; step[nx + tid] = un[tid]; S.du0[tid] = S.hot[L.oDumin + tid] + un[tid]; S.du0[nu + tid] = S.hot[L.oDumax + tid] + un[tid]; }
            if (R.xref_traj) for (int i = tid; i < R.xref_blk; i += NT) { const double xr = R.xref_traj[kb * R.xref_blk + i]; step[nx + nu + i] = xr; if (i < nx) S.xrs[i] = xr; }
            __syncthreads();
        }
#ifdef MPCQP_RUN_TIMING
#define PHASE_CLOCK(i) { unsigned long long t_ = wall_clock64(); if (tid == 0) atomicAdd(&P.stats[4 + (i)], t_ - tphase); tphase = t_; }
        unsigned long long tphase = wall_clock64();
#else
#define PHASE_CLOCK(i)
#endif
        int iter = 0, term = 0;
        if (!LOOP && R.part == 2) {                           // resumed: the first round and its check are done
            const mpcqp_info pi = P.info[b];
            iter = pi.iter;
            if (tid == 0) { S.iflag[1] = pi.rho_updates; S.iflag[3] = pi.reserved; }
        }
        else { FramePin pin; run_begin_phase<NB, LDSSTATE, OCC>(R.plain, (R.warm_x && k == 0) ? 1 : 0, &pin.v); }
        __syncthreads();
        PHASE_CLOCK(0)
        while (!term) {
            const int nxt = next_stop(iter, R.max_iter, R.chk, R.rho_every);
            term = __builtin_amdgcn_readfirstlane(run_admm<NB, LDSSTATE, NXT, NUT, MODE, OCC>(nxt - iter, R.plain ? -1 : iter));
            iter = nxt;
            __syncthreads();
            if constexpr (MODE >= MODE_BCRT) { if (!R.plain) iter = S.iflag[5]; }      // (the latency round may have run several rounds, and finished the solve: admm_latw)
            PHASE_CLOCK(1)
            if (term) break;
            { FramePin pin; term = run_check_phase<NB, LDSSTATE, OCC>(iter, stop_mode(iter, R.max_iter, R.chk, R.rho_every, R.plain != 0), &pin.v); }
            __syncthreads();
            PHASE_CLOCK(2)
            if (!LOOP && R.part == 1 && !term) {              // hand the instance over to the follow-up launch
                if (tid == 0) R.pending[atomicAdd(R.npending, 1)] = b;
                break;
            }
        }
        if (LOOP && tid == 0) {
            R.status_traj[(size_t)k * R.batch + b] = S.iflag[4];
            R.iter_traj[(size_t)k * R.batch + b] = iter;
            if (k < TS_STEPS) P.tstamp[(size_t)TS_STRIDE * b + 2 + k] = wall_clock64();
        }
        __syncthreads();
    }
    if (!LOOP && R.pub) {                        // results to the caller's (mapped) memory; the last workgroup raises the flag
        double *px = R.pub + (size_t)b * L.n, *py = R.pub + (size_t)R.batch * L.n + (size_t)b * L.m;
        for (int j = tid; j < L.n; j += NT) px[j] = P.xo[(size_t)b * L.n + j];
        for (int r = tid; r < L.m; r += NT) py[r] = P.yo[(size_t)b * L.m + r];
        mpcqp_info *pi = (mpcqp_info *)(R.pub + (size_t)R.batch * (L.n + L.m));
        if (tid == 0) pi[b] = P.info[b];
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            if (atomicAdd(R.done, 1u) == (unsigned)R.batch - 1u) {      // (one workgroup per instance on this path: never persistent)
                *R.done = 0;
                __threadfence_system();
                *(volatile unsigned long long *)(pi + R.batch) = R.seq;
            }
        }
    }
    if (LOOP && k1 < R.nsteps) return;           // (a queue item that is not the instance's last)
    if (LOOP && tid == 0) P.tstamp[(size_t)TS_STRIDE * b + 1] = wall_clock64();
    if (LOOP && tid < nx) {
        const size_t e = ((size_t)R.nsteps * R.batch + b) * nx + tid;
        R.x_traj[e] = R.ny ? R.x_true[(size_t)b * nx + tid] : S.x0s[tid];
        if (R.ny && R.xhat_traj) R.xhat_traj[e] = S.x0s[tid];
    }
}

// The kernel.  Grid = batch: one workgroup per instance (inst_of: the workgroup -> instance map).  PERSISTENT (RunArgs::vcur, batches beyond the
// resident slots): the grid is the number of resident slots and every workgroup takes ITEMS off a queue until it is empty.  An item is an instance --
// position v of the map is instance vperm[v], the map is the longest-expected-work-first list (rebalance() in mpcqp.hip): the greedy schedule -- or, in
// the closed loop, a few consecutive steps of an instance (long parts first, short ones last): the items of part p + 1 follow all items of part p in the queue, an item waits for its
// instance's previous part through a per-instance progress counter (vdone; its producer was taken off the queue earlier and never waits for a later
// item, so this cannot deadlock), and the launch ends within a fraction of an instance's closed loop of its ideal length instead of within a whole one.
// Measured on the hardware's own dispatch of 1024 one-at-a-time workgroups (scripts/diag_makespan.py, HW_ID stamps): every compute unit gets exactly
// four of them whatever they last, 54 us (median; mean 116) pass between two workgroups on a compute unit, and the units idle 1.3 ms of a 12.6 ms
// launch at its end -- 13 % of the slots' time.  The instance a workgroup works on reaches the phases the way it always did, through inst_of(P.perm):
// P.perm points at vcur, one entry per workgroup, which thread 0 rewrites per item (fence, barrier, scalar-cache invalidate: the phases read it with
// scalar loads).
template <int NB, bool LDSSTATE, int NXT, int NUT, int MODE, bool LOOP>
__global__ __launch_bounds__(NT, run_occupancy(NB, MODE)) void k_mpc_run(RunKArgs A_) {
    const RunArgs &R = run_kargs().R;
    const int nrun = LOOP ? R.nsteps : 1;
    if (R.vcur == nullptr) {
        if (!LOOP && R.part == 2 && (int)blockIdx.x >= *R.npending) return;
        run_instance<NB, LDSSTATE, NXT, NUT, MODE, LOOP>(0, nrun);
        return;
    }
    __shared__ int s_item[3];                                  // map entry (-1: the queue is empty), first step, end step
    const int nvirt = (!LOOP && R.part == 2) ? *R.npending : R.batch;
    const int nparts = (LOOP && R.vparts > 1) ? R.vparts : 1;
    for (;;) {
        if (threadIdx.x == 0) {
            const int v = (int)atomicAdd(R.vqueue, 1u);
            int entry = -1, ka = 0, kb = nrun;
            if (v < nvirt * nparts) {
                const int part = v / nvirt, idx = v - part * nvirt;
                entry = R.vperm ? R.vperm[idx] : idx;
                if (nparts > 1) { ka = R.voff[part]; kb = R.voff[part + 1]; }
                if (part > 0) {                                // the instance's previous part must be done (and its writes visible: acquire below)
                    volatile int *done = R.vdone + (entry & PERM_INST_MASK);
                    while (*done < ka) __builtin_amdgcn_s_sleep(16);
                }
                R.vcur[blockIdx.x] = entry;                   // (read by this workgroup only: the barrier below orders it)
            }
            s_item[0] = entry; s_item[1] = ka; s_item[2] = kb;
        }
        __syncthreads();
        const int entry = s_item[0], ka = s_item[1], kb = s_item[2];
        if (entry < 0) {
            // the last workgroup to leave puts the queue back to zero for the next launch (vqueue[1] counts the leavers): no memset between launches
            if (threadIdx.x == 0 && atomicAdd(R.vqueue + 1, 1u) == gridDim.x - 1) { R.vqueue[1] = 0; __threadfence(); R.vqueue[0] = 0; }
            break;
        }
        __builtin_amdgcn_s_dcache_inv();                       // (inst_of's load of vcur[blockIdx.x] is a scalar load in most phases)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // (... and a vector load in the others; with parts: everything the previous part wrote)
        run_instance<NB, LDSSTATE, NXT, NUT, MODE, LOOP>(ka, kb);
        __syncthreads();
        if (nparts > 1 && threadIdx.x == 0) { __threadfence(); *(volatile int *)(R.vdone + (entry & PERM_INST_MASK)) = kb; }
    }
}

// ------------------------------------------------------------------------------------------------
// verification kernels
// ------------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(NT) void k_export(Lay L, Ptrs P, double *Pd, double *Ad_, double *q, double *l, double *u) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *p = sh; Smem S; smem_common(L, P, p, S);
    const int b = inst_of(P.perm), tid = threadIdx.x;
    const double *model = P.model + (size_t)b * L.model_sz, *step = P.step + (size_t)b * L.step_sz;
    load_common(L, model, step, S);
    Ctx c{L, S.hot, model + L.hot_sz};
    if (!L.raw) build_q(c, step, S.Qv);
    __syncthreads();
    if (Pd) { double *o = Pd + (size_t)b * L.n * L.n; for (int j = tid; j < L.n; j += NT) P_row(c, j, [&](double co, int idx) { o[(size_t)j * L.n + idx] = co; }); }
    if (Ad_) { double *o = Ad_ + (size_t)b * L.m * L.n; for (int r = tid; r < L.m; r += NT) A_row(c, r, [&](double co, int idx) { o[(size_t)r * L.n + idx] = co; }); }
    if (q) for (int j = tid; j < L.n; j += NT) q[(size_t)b * L.n + j] = (j < L.oe) ? S.Qv[j] : 0.0;
    if (l && u) for (int r = tid; r < L.m; r += NT) { double lo, hi; row_bounds(c, S.x0s, S.du0, r, lo, hi); l[(size_t)b * L.m + r] = lo; u[(size_t)b * L.m + r] = hi; }
}

template <int NB>
__global__ __launch_bounds__(NT) void k_kkt_solve(Lay L, Ptrs P, const double *rhs, double *sol) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *p = sh; Smem S; smem_common(L, P, p, S);
    const int b = inst_of(P.perm), tid = threadIdx.x;
    const double *model = P.model + (size_t)b * L.model_sz, *step = P.step + (size_t)b * L.step_sz;
    load_common(L, model, step, S);
    Ctx c{L, S.hot, model + L.hot_sz};
    kkt_solve<NB>(c, P.omega + (size_t)b * L.m, P.s + (size_t)b * L.n, P.c[b], factor_of(P, b), rhs + (size_t)b * L.n, S.T + L.m, sol + (size_t)b * L.n, border_ptrs(L, P, S), S.tv);
    (void)tid;
}

// ------------------------------------------------------------------------------------------------
// mpcqp_eq_solve: the EQUALITY-constrained QP (the dynamics rows alone; every other row ignored) by the method of multipliers in
// RESIDUAL form, preconditioned by the handle's KKT factor:
//     r  = -c (P x + q + A_e' y) - A_e' ( omega_e . (A_e x - b) )        (exact operators: the row visitors of mpcqp_qp.h)
//     x += K^-1 r                                                         (kkt_solve: whatever backend the handle has)
//     y += (omega_e / c) (A_e x - b)
// K = c P + diag(s) + A' diag(omega) A is what the ADMM iteration solves with, so in exact arithmetic a sweep IS an ADMM iteration with
// alpha = 1 on a problem whose other rows are free; written with the residual, the fixed point is the exact KKT point whatever the
// rounding error of the stored factor (which only slows the contraction), so rho can be large and a sweep contracts by ~ |P| / rho_eq.
// This is what the unconstrained gains of test_scripts/alternative/unconstrained.py:170-183 need: one factorization, a few solves.
// At most `sweeps` sweeps; an instance stops earlier once a correction is below tol * max(1, |x|_inf) (tol = 0: never).
// res [batch][5]: |P x + q + A_e'y|_inf, max(|P x|, |A_e'y|, |q|)_inf, |A_e x - b|_inf, max(|A_e x|, |b|)_inf after the last sweep, sweeps done.
// Plain loops over the visitors: a utility on a handful of instances, not a hot path.
// ------------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(NT) void k_eq_solve(Lay L, Ptrs P, int sweeps, int cold, double tol, double *res) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *p = sh; Smem S; smem_common(L, P, p, S);
    const int b = inst_of(P.perm), tid = threadIdx.x;
    const double *model = P.model + (size_t)b * L.model_sz, *step = P.step + (size_t)b * L.step_sz;
    load_common(L, model, step, S);
    Ctx c{L, S.hot, model + L.hot_sz};
    if (!L.raw) build_q(c, step, S.Qv);
    double *x = P.x + (size_t)b * L.n, *y = P.y + (size_t)b * L.m, *z = P.z + (size_t)b * L.m;
    double *r = P.dx + (size_t)b * L.n, *d = P.xo + (size_t)b * L.n;
    const double *om = P.omega + (size_t)b * L.m, *sv = P.s + (size_t)b * L.n;
    const double cc = P.c[b];
    double *g = S.T;                                   // [n_x]: c y + omega (A x - b) of the dynamics rows (W's place in the work vector)
    if (cold) { for (int j = tid; j < L.n; j += NT) x[j] = 0.0; for (int i = tid; i < L.m; i += NT) { y[i] = 0.0; z[i] = 0.0; } }
    __syncthreads();
    double nrm[4] = {0.0, 0.0, 0.0, 0.0}, dummy[1] = {0.0};
    int done = 0;
    bool settled = false;
    bool broken = P.info[b].status == MPCQP_NON_CVX && P.info[b].iter == 0;      // (k_setup's verdict on its factorization: a non-positive pivot)
    for (int sw = 0; sw <= sweeps && !broken; ++sw) {
        nrm[0] = nrm[1] = nrm[2] = nrm[3] = 0.0;
        for (int i = tid; i < L.n_x; i += NT) {
            double ax = 0.0, lo, hi;
            A_row(c, i, [&](double co, int idx) { ax += co * x[idx]; });
            row_bounds(c, S.x0s, S.du0, i, lo, hi);
            g[i] = cc * y[i] + om[i] * (ax - lo);
            nrm[2] = fmax(nrm[2], fabs(ax - lo)); nrm[3] = fmax(nrm[3], fmax(fabs(ax), fabs(lo)));
        }
        __syncthreads();
        for (int j = tid; j < L.n; j += NT) {
            double px = 0.0, atg = 0.0, aty = 0.0;
            P_row(c, j, [&](double co, int idx) { px += co * x[idx]; });
            AT_row(c, j, [&](double co, int row) { if (row < L.n_x) { atg += co * g[row]; aty += co * y[row]; } });
            const double qj = (j < L.oe) ? S.Qv[j] : 0.0;
            r[j] = -cc * (px + qj) - atg;
            nrm[0] = fmax(nrm[0], fabs(px + qj + aty)); nrm[1] = fmax(nrm[1], fmax(fabs(px), fmax(fabs(aty), fabs(qj))));
        }
        __syncthreads();
        if (sw == sweeps || settled) break;
        kkt_solve<NB>(c, om, sv, cc, factor_of(P, b), r, S.T + L.m, d, border_ptrs(L, P, S), S.tv);
        double mx[2] = {0.0, 0.0}, dsum[1] = {0.0};
        for (int j = tid; j < L.n; j += NT) { const double dj = d[j], xj = x[j] + dj; x[j] = xj; mx[0] = fmax(mx[0], fabs(dj)); mx[1] = fmax(mx[1], fabs(xj)); dsum[0] += dj; }
        block_reduce<2, 1>(mx, dsum, S.red);
        done = sw + 1;
        // (fmax drops NaN operands: a NaN correction -- a bad factor, a non-positive pivot -- would leave mx[0] = 0 and read as 'settled';
        //  the plain sum of the corrections carries it, as the objective does in check_body)
        if (dsum[0] != dsum[0]) { broken = true; break; }
        settled = mx[0] <= tol * fmax(1.0, mx[1]);       // (the residuals of the settled iterate are evaluated by one more pass of the loop head)
        for (int i = tid; i < L.n_x; i += NT) {
            double ax = 0.0, lo, hi;
            A_row(c, i, [&](double co, int idx) { ax += co * x[idx]; });
            row_bounds(c, S.x0s, S.du0, i, lo, hi);
            y[i] += (om[i] / cc) * (ax - lo);
        }
        __syncthreads();
    }
    block_reduce<4, 1>(nrm, dummy, S.red);
    if (tid < 4) res[(size_t)b * 5 + tid] = nrm[tid];
    if (tid == 4) res[(size_t)b * 5 + 4] = (double)done;
    // the result as a solve reports it: solution, iterate (z = A x: what a warm start would begin from), info
    double *xo = P.xo + (size_t)b * L.n, *yo = P.yo + (size_t)b * L.m;
    for (int j = tid; j < L.n; j += NT) xo[j] = x[j];
    for (int i = tid; i < L.m; i += NT) {
        if (i >= L.n_x) y[i] = 0.0;
        yo[i] = y[i];
        double ax = 0.0; A_row(c, i, [&](double co, int idx) { ax += co * x[idx]; });
        z[i] = ax;
    }
    if (tid == 0) {
        // 'solved': the corrections have vanished (or no tolerance was asked for); 'maximum iterations reached': `sweeps` sweeps did not settle it
        // 'problem non convex' (OSQP's word for a factorization that failed): a NaN correction, or a factor k_setup already reported bad
        mpcqp_info inf; inf.status = broken ? MPCQP_NON_CVX : (tol > 0.0 && !settled) ? MPCQP_MAX_ITER_REACHED : MPCQP_SOLVED; inf.iter = done; inf.rho_updates = 0; inf.reserved = 0;
        inf.obj_val = broken ? NAN : 0.0; inf.pri_res = broken ? NAN : nrm[2]; inf.dua_res = broken ? NAN : nrm[0]; inf.rho = P.rho[b];
        P.info[b] = inf;
    }
}

__global__ void k_gather_u0(Lay L, const double *xo, double *u0, int batch) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < batch * L.nu) { int b = i / L.nu, j = i - b * L.nu; u0[i] = xo[(size_t)b * L.n + L.ou + j]; }
}

// mpcqp_update_vectors: raw q, l, u of the reference's layout (mpc.py:593-599) -> the tables the kernels work from.
// q: the x,u part verbatim (the slack part of the reference's q is identically zero).  l, u: row blocks are stage-periodic
// in the reference (mpc.py:551-580), so one period is stored: x0 = -l[:nx]; state box = soft rows of stage 0; input box =
// rows of u_0; Delta-u bounds = the rows behind the first nu (which carry Dumin/Dumax + u_{-1} and are kept verbatim).
__global__ void k_decode_vectors(Lay L, Ptrs P, const double *q, const double *l, const double *u, int batch) {
    const int b = blockIdx.x, tid = threadIdx.x;
    double *model = P.model + (size_t)b * L.model_sz, *step = P.step + (size_t)b * L.step_sz;
    if (q) for (int j = tid; j < L.n_x + L.n_u; j += blockDim.x) P.qv[(size_t)b * (L.n_x + L.n_u) + j] = q[(size_t)b * L.n + j];
    const double *lb = l ? l + (size_t)b * L.m : nullptr, *ub = u ? u + (size_t)b * L.m : nullptr;
    for (int i = tid; i < L.nx; i += blockDim.x) {
        if (lb) { step[i] = -lb[i]; model[L.oxmin + i] = lb[L.rs + i]; } else if (ub) step[i] = -ub[i];
        if (ub) model[L.oxmax + i] = ub[L.rs + i];
    }
    for (int j = tid; j < L.nu; j += blockDim.x) {
        if (lb) { model[L.oumin + j] = lb[L.ri + j]; model[L.oDumin + j] = lb[L.rdu + L.nu + j]; step[L.odu0 + j] = lb[L.rdu + j]; }
        if (ub) { model[L.oumax + j] = ub[L.ri + j]; model[L.oDumax + j] = ub[L.rdu + L.nu + j]; step[L.odu0 + L.nu + j] = ub[L.rdu + j]; }
    }
    (void)batch;
}

// entering raw-vector mode with only some of q, l, u given: the first Delta-u rows keep their current bounds Dumin/Dumax + u_{-1}
__global__ void k_keep_du0(Lay L, Ptrs P, int batch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < batch * L.nu) {
        const int b = i / L.nu, j = i - b * L.nu;
        const double *model = P.model + (size_t)b * L.model_sz; double *step = P.step + (size_t)b * L.step_sz;
        step[L.odu0 + j] = model[L.oDumin + j] + step[L.nx + j];
        step[L.odu0 + L.nu + j] = model[L.oDumax + j] + step[L.nx + j];
    }
}

// output() of mpc.py:271-336 for the whole batch: u = first input of the solution if the status is 'solved', else
// u_failure (= uref); optionally also becomes u_{-1} of the next update (output() sets uminus1_rh).
__global__ void k_output_u(Lay L, Ptrs P, double *u_out, int batch, int store_um1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < batch * L.nu) {
        const int b = i / L.nu, j = i - b * L.nu;
        const double u = P.info[b].status == MPCQP_SOLVED ? P.xo[(size_t)b * L.n + L.ou + j] : P.model[(size_t)b * L.model_sz + L.ouref + j];
        u_out[i] = u;
        if (store_um1) P.step[(size_t)b * L.step_sz + L.nx + j] = u;
    }
}
