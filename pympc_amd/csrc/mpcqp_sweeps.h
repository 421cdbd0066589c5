// mpcqp_sweeps.h -- part of libmpcqp_hip (included by mpcqp.hip, one translation unit).
// The solve with the factor: forward elimination from the forward matrices (chain_sweep), the sweeps that apply the
// off-diagonal blocks matrix-free (so_sweep), the twisted solve built from them (kkt_core).
#pragma once

// The sweeps work on Tc: the x,u part of the right-hand side / solution in STAGE-MAJOR PADDED layout,
// Tc[k*NB + a] = element a of stage k (a < nx: x_k[a]; nx <= a < nb: u_k[a-nx]; everything else is padding
// and stays exactly zero because the stored S^-1 has zero rows there).  In the operand layout lane 16k + 4b + j
// holds element 4b + k (+16 per block): one 8-byte LDS read per lane and block.
template <int NB>
__device__ __forceinline__ void vec_load(const double *tb, int k, double *v) {
#pragma unroll
    for (int bi = 0; bi < NB / 16; ++bi) v[bi] = tb[k * NB + bi * 16];
}
template <int NB>
__device__ __forceinline__ void vec_store(double *tb, int k, const double *v, bool writer) {
    if (writer) {
#pragma unroll
        for (int bi = 0; bi < NB / 16; ++bi) tb[k * NB + bi * 16] = v[bi];
    }
}
// per-lane base of a stage vector in Tc: lane 16k + 4b + j holds element 4b + k
__device__ __forceinline__ int vec_lane_offset(int lane) { return 4 * ((lane >> 2) & 3) + (lane >> 4); }
// A stage vector is replicated over the four lanes j of a row position; one of them writes it back -- or, inside the sweeps,
// all four (MPCQP_STORE_ALL: the same value to the same address, no exec-mask juggling and no extra basic block per stage).
#ifndef MPCQP_STORE_ALL
#define MPCQP_STORE_ALL 1
#endif
__device__ __forceinline__ bool vec_lane_writer(int lane) { return (lane & 3) == 0; }

// rotate every 16-lane row by 4*sft lanes: lane (k, b, j) receives the value of lane (k, (b+sft)%4, j)
// (`old` operand of the DPP move: with row_ror every lane receives a value, so what the destination held before is
//  irrelevant -- an unspecified register instead of a zero saves the two v_mov_b32 the compiler would otherwise emit per
//  rotation, ~500 instructions per workgroup and iteration in the sweeps)
#define DPP_MOVE(src, ctrl) __builtin_amdgcn_mov_dpp((src), (ctrl), 0xF, 0xF, false)      // (v_mov_b32_dpp with an undefined `old`)
template <int SFT>
__device__ __forceinline__ double rot_blocks(double x) {
    if (SFT == 0) return x;
    constexpr int CTRL = 0x120 | (16 - 4 * SFT);          // row_ror:n gives dst[i] = src[(i - n) mod 16]
    const long long xi = __builtin_bit_cast(long long, x);
    const int lo = DPP_MOVE((int)xi, CTRL);
    const int hi = DPP_MOVE((int)(xi >> 32), CTRL);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// out[bi] += sum_bj A(bi,bj) * in[bj]   with A given as fragments (one d4 per block per lane)
// The four MFMAs of a block are issued as TWO dependent pairs whose partial sums are added on the vector ALU: a stage of a
// sweep is a latency chain, and a dependent f64 4x4x4 MFMA costs ~44 cycles -- two in a row plus one add instead of four.
#ifndef MPCQP_MFMA_PAIRS
#define MPCQP_MFMA_PAIRS 1
#endif
template <int NB>
__device__ __forceinline__ void frag_matvec(const d4 *A, const double *in, double *out) {
    constexpr int NBLK = NB / 16;
#if MPCQP_MFMA_PAIRS
    double side[NBLK];
#pragma unroll
    for (int bi = 0; bi < NBLK; ++bi) side[bi] = 0.0;
#endif
#pragma unroll
    for (int bj = 0; bj < NBLK; ++bj) {
        const double r0 = in[bj], r1 = rot_blocks<1>(in[bj]), r2 = rot_blocks<2>(in[bj]), r3 = rot_blocks<3>(in[bj]);
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi) {
            const d4 a = A[bi * NBLK + bj];
#if MPCQP_MFMA_PAIRS
            out[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], r0, out[bi], 0, 0, 0);
            side[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[2], r2, side[bi], 0, 0, 0);
            out[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], r1, out[bi], 0, 0, 0);
            side[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[3], r3, side[bi], 0, 0, 0);
#else
            out[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], r0, out[bi], 0, 0, 0);
            out[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], r1, out[bi], 0, 0, 0);
            out[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[2], r2, out[bi], 0, 0, 0);
            out[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[3], r3, out[bi], 0, 0, 0);
#endif
        }
    }
#if MPCQP_MFMA_PAIRS
#pragma unroll
    for (int bi = 0; bi < NBLK; ++bi) out[bi] += side[bi];
#endif
}

// The 32 x 32 sweeps read the factor stream through BUFFER loads: resource = the instance's factor, scalar offset = the stage,
// vector offset = the lane's constant byte offset inside a stage record -- the address of a load needs no vector arithmetic at
// all (a global load takes a 64-bit per-lane address, formed with one or two VALU instructions per load and stage: ten of the
// ~150 instructions of a stage).  cfg-5 +1 %; at 16 x 16 (three loads per stage) it measured -0.7 %, so those keep global loads.
#ifndef MPCQP_BUFFER_LOADS
#define MPCQP_BUFFER_LOADS 1
#endif
typedef unsigned int bu4 __attribute__((ext_vector_type(4)));
typedef unsigned int bu2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t factor_rsrc(const double *F) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)F, 0, 0x7fffffff, 0x00020000);       // raw buffer, no range check worth having
}
__device__ __forceinline__ double bu_double(unsigned lo, unsigned hi) { return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo); }
__device__ __forceinline__ d4 buf_load_d4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const bu4 a = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0), b = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16, soff, 0);
    return d4{bu_double(a[0], a[1]), bu_double(a[2], a[3]), bu_double(b[0], b[1]), bu_double(b[2], b[3])};
}
template <int NB>
__device__ __forceinline__ void frag_load(const double *Fm, int lane, d4 *A) {
    constexpr int NBLK = NB / 16;
#pragma unroll
    for (int b = 0; b < NBLK * NBLK; ++b) A[b] = *(cgd4 *)(Fm + b * 256 + lane * 4);
}
// S_k^-1 is symmetric: of a 16 x 16 block only the 4x4 blocks on and above the block diagonal are stored -- 160 of 256 doubles,
// packed per operand-layout lane: a lane of block row R owns its 4-R values (steps 0..3-R) back to back,
//     offset(lane = 16k + 4R + i) = 40 k + cum(R) + i (4-R),   cum = 0, 16, 28, 36      (+4 doubles of slack per fragment)
// A lane loads a 4-double window at its offset (the tail of the window belongs to the next lane and is discarded) and the
// missing steps are rebuilt from the transposed block: step s of block row R with R+s >= 4 is block (R, R+s-4) =
// block (R+s-4, R)', i.e. step 4-s of lane 16 i + 4 (R+s-4) + k -- three cross-lane permutes (sym_expand16).
__device__ __forceinline__ double lane_permute(double x, int byte_addr) {
    const long long xi = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, (int)xi), hi = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(xi >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
template <int NB> struct SweepCfg {
    static constexpr int NF = (NB / 16) * (NB / 16);
#ifndef MPCQP_DEPTH
#define MPCQP_DEPTH 4
#endif
#ifndef MPCQP_DEPTH32
#define MPCQP_DEPTH32 4
#endif
    static constexpr int DEPTH = NB == 32 ? MPCQP_DEPTH32 : MPCQP_DEPTH;   // factor stages kept in flight in registers (8 VGPRs each at NB = 16, 32 at NB = 32)
};

// The sweeping waves are dependent MFMA chains: two of them on one SIMD share its matrix pipe and slow each other
// down.  Workgroups that are co-resident on a CU (dispatch order: block b -> XCD b%8, CU (b/8)%32) therefore rotate
// which of their waves does what, so that the sweepers of the four co-resident workgroups spread over the four
// SIMDs.  Purely a speed matter: any placement gives the same results.
__device__ __forceinline__ int logical_wave() {
#ifdef MPCQP_NO_WAVE_ROTATION
    return threadIdx.x >> 6;
#else
    return ((threadIdx.x >> 6) - (blockIdx.x >> 8)) & (NWAVES - 1);
#endif
}

// Forward elimination of one half-chain by ONE wave: for i = 1..nsteps, k = first + dir*i:   Tc[k] <- Tc[k] + Fwd(k) * Tc[k - dir]
// (Fwd(k) = the forward-matrix slot of stage k, which holds the negated factor block).  The factor fragments of the next DEPTH
// stages are prefetched into a register ring; the running vector ping-pongs between two register sets (no copies between MFMAs).
template <int NB>
__device__ __forceinline__ void chain_sweep(const int first, const int dir, const int nsteps, const int fstage, const double *F, double *Tc) {
    constexpr int NBLK = NB / 16, NF = SweepCfg<NB>::NF, DEPTH = SweepCfg<NB>::DEPTH;
    const int lane = opaque_lane(threadIdx.x & 63);
    double *tb = Tc + vec_lane_offset(lane);
    const bool writer = MPCQP_STORE_ALL ? true : vec_lane_writer(lane);
    auto stage_of = [&](int i) { return first + dir * i; };
    auto frag_of = [&](int i) { return F + (size_t)stage_of(i) * fstage; };
    // The group loop below is branch-free on purpose: with conditionals around the refills the compiler can no longer
    // count the loads in flight across the back edge and falls back to s_waitcnt vmcnt(0) -- the whole memory latency
    // once per group.  Refills past the end re-read the last stage (clamped index), the tail group runs separately.
    auto frag_clamped = [&](int i) { return frag_of(i < nsteps ? i : nsteps); };
    auto ring_load = [&](int i, d4 *A) { frag_load<NB>(frag_clamped(i), lane, A); };
    d4 ring[DEPTH][NF];
    if (nsteps < 1) return;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ring_load(1 + d, ring[d]);
    double va[NBLK], vb[NBLK];
    vec_load<NB>(tb, first, va);
    // The stage's own vector is read from LDS ONE STAGE AHEAD (oa / ob alternate): read where it is used, its LDS round trip
    // sits on the dependent chain of every stage.  (Stage k+dir is not written before its own step.)
    double oa[NBLK], ob[NBLK];
    vec_load<NB>(tb, stage_of(1), oa);
    auto stage_step = [&](int i, int d) {
        const int k = stage_of(i);
        double *src = (d & 1) ? vb : va, *dst = (d & 1) ? va : vb;
        double *own = (d & 1) ? ob : oa, *nxt = (d & 1) ? oa : ob;
        vec_load<NB>(tb, stage_of(i < nsteps ? i + 1 : i), nxt);
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi) dst[bi] = own[bi];
        frag_matvec<NB>(ring[d], src, dst);
        vec_store<NB>(tb, k, dst, writer);
    };
    int i0 = 1;
    for (; i0 + DEPTH - 1 <= nsteps; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            stage_step(i0 + d, d);
            ring_load(i0 + d + DEPTH, ring[d]);
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (i0 + d <= nsteps) stage_step(i0 + d, d);
}

// What the linear-system core needs to know about one instance.
// F: the stages; G: the constant fragments [G | G'] (in front of the stages in the S^-1-only format, behind them otherwise).
struct CoreArgs { int N, fstage, nx, nu, NcT, rdu, grp; const double *F; const double *G; const double *om; };
__device__ __forceinline__ CoreArgs core_args(const Lay &L, const double *F, const double *om) {
    CoreArgs a; a.N = L.N; a.fstage = L.fstage; a.nx = L.nx; a.nu = L.nu; a.NcT = L.NcT; a.rdu = L.rdu; a.om = om; a.grp = L.grp;
    a.F = L.ffwd ? F : F + L.fhead;
    a.G = L.ffwd ? F + (size_t)L.N * L.fstage : F;
    return a;
}

// ------------------------------------------------------------------------------------------------
// S^-1-only factor format (FactorFmt<NB>::SONLY): the twisted solve with the off-diagonal blocks applied matrix-free.
//     forward   w_k = S_k^-1 ( b_k - K_{k,nbr} w_nbr )            top half: nbr = k-1, bottom half: nbr = k+1
//     middle    x_m = S_m^-1 ( b_m - K_{m,m-1} w_{m-1} - K_{m,m+1} w_{m+1} )
//     backward  x_k = w_k - S_k^-1 K_{k,nbr} x_nbr                top half: nbr = k+1, bottom half: nbr = k-1
// With G = [[Ad, Bd], [0, c QDu']] (constant per instance, NB x NB, zero padded), om_s = omega of the dynamics rows of
// stage s (x lanes; 1 on the others) and wd_s = omega of the Delta-u row that couples u_s[nu-1] with u_{s+1}[0] (mpc.py:570):
//     -K_{k,k-1} v = mask_k . ( om_k . (G v) )      + e_{nx}      wd_{k-1} v[nx+nu-1]          (neighbour above: "up")
//     -K_{k,k+1} v = mask_k . ( G' (om_{k+1} . v) ) + e_{nx+nu-1} wd_k     v[nx]               (neighbour below)
// (kkt_sub_entry in mpcqp_qp.h is the entry-wise definition; mask_k clears the u rows of stages that carry no input and
// the padding.)  A stage streams its packed S^-1 and 2 omega values per lane; G (top half forward, bottom half backward)
// or G' sits in registers for the length of a sweep.
// ------------------------------------------------------------------------------------------------
template <int NB> struct SoCfg {
    static constexpr int NBLK = NB / 16, NF = NBLK * NBLK;
    static constexpr int NS = NB == 16 ? 1 : 3;               // streamed d4 per lane and stage: sym(S) | sym(S00), sym(S11), S01
#ifndef MPCQP_SO_DEPTH16
#define MPCQP_SO_DEPTH16 3
#endif
#ifndef MPCQP_SO_DEPTH32
#define MPCQP_SO_DEPTH32 2
#endif
    static constexpr int DEPTH = NB == 32 ? MPCQP_SO_DEPTH32 : MPCQP_SO_DEPTH16;      // stages in flight (12 VGPRs each at NB = 16, 40 at NB = 32)
};
__device__ __forceinline__ d4 sym_window(const double *Fm, int lane) {
    const int R = (lane >> 2) & 3;
    const d4u w = *(cgd4u *)(Fm + 40 * (lane >> 4) + sym_cum(R) + (lane & 3) * (4 - R));
    return d4{w[0], w[1], w[2], w[3]};
}
__device__ __forceinline__ d4 sym_expand16(const d4 w, int lane) {
    const int R = (lane >> 2) & 3, k = lane >> 4, i = lane & 3;
    const double t1 = lane_permute(w[3], 4 * (16 * i + 4 * ((R + 1) & 3) + k));
    const double t2 = lane_permute(w[2], 4 * (16 * i + 4 * ((R + 2) & 3) + k));
    const double t3 = lane_permute(w[1], 4 * (16 * i + 4 * ((R + 3) & 3) + k));
    return d4{w[0], R + 1 >= 4 ? t1 : w[1], R + 2 >= 4 ? t2 : w[2], R + 3 >= 4 ? t3 : w[3]};
}
// fragment of the transposed 16 x 16 block: element (r, c) of B' = element (c, r) of B, i.e. step s of lane (k, b, i) is
// step (4 - s) & 3 of lane (i, (b + s) & 3, k)
__device__ __forceinline__ d4 frag_transpose16(const d4 w, int lane) {
    const int b = (lane >> 2) & 3, k = lane >> 4, i = lane & 3;
    return d4{lane_permute(w[0], 4 * (16 * i + 4 * b + k)), lane_permute(w[3], 4 * (16 * i + 4 * ((b + 1) & 3) + k)),
              lane_permute(w[2], 4 * (16 * i + 4 * ((b + 2) & 3) + k)), lane_permute(w[1], 4 * (16 * i + 4 * ((b + 3) & 3) + k))};
}
template <int NB>
__device__ __forceinline__ void so_load(const double *Fk, int lane, d4 *A) {
    if constexpr (NB == 16) A[0] = sym_window(Fk, lane);
    else { A[0] = sym_window(Fk, lane); A[1] = sym_window(Fk + 164, lane); A[2] = *(cgd4 *)(Fk + 328 + lane * 4); }
}
template <int NB>
__device__ __forceinline__ void so_expand(const d4 *A, int lane, d4 *Sf) {
    if constexpr (NB == 16) Sf[0] = sym_expand16(A[0], lane);
    else { Sf[0] = sym_expand16(A[0], lane); Sf[3] = sym_expand16(A[1], lane); Sf[1] = A[2]; Sf[2] = frag_transpose16(A[2], lane); }
}
__device__ __forceinline__ double lane_bcast(double x, int src) {
    const long long xi = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readlane((int)xi, src), hi = __builtin_amdgcn_readlane((int)(xi >> 32), src);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
template <int NB> struct SoLane {                 // per-lane constants of the operand layout
    int e[NB / 16];                               // own element per block
    bool isx[NB / 16], isu[NB / 16];
};
template <int NB>
__device__ __forceinline__ SoLane<NB> so_lane(const CoreArgs &a, int lane) {
    SoLane<NB> q;
#pragma unroll
    for (int bi = 0; bi < NB / 16; ++bi) {
        q.e[bi] = 16 * bi + vec_lane_offset(lane);
        q.isx[bi] = q.e[bi] < a.nx; q.isu[bi] = q.e[bi] >= a.nx && q.e[bi] < a.nx + a.nu;
    }
    return q;
}
// value of element E of a stage vector held one double per lane and block (uniform E)
template <int NB>
__device__ __forceinline__ double so_element(const double *v, int E) {
    const int l = E & 15, src = 16 * (l & 3) + 4 * (l >> 2);
    if constexpr (NB == 16) return lane_bcast(v[0], src);
    else { const double a0 = lane_bcast(v[0], src), a1 = lane_bcast(v[1], src); return (E >> 4) ? a1 : a0; }
}
// t = -K_{k,nbr} v   (UP: nbr = k-1, uses G; else nbr = k+1, uses G').  sc = om of stage max(k,nbr) on x lanes (1 elsewhere).
template <int NB, bool UP>
__device__ __forceinline__ void so_offdiag(const CoreArgs &a, const SoLane<NB> &q, const d4 *Gf, int k, const double *v,
                                           const double *sc, double wd, double *t) {
    constexpr int NBLK = NB / 16;
    double in[NBLK], out[NBLK];
#pragma unroll
    for (int bi = 0; bi < NBLK; ++bi) { in[bi] = UP ? v[bi] : sc[bi] * v[bi]; out[bi] = 0.0; }
    frag_matvec<NB>(Gf, in, out);
    const int Esrc = UP ? a.nx + a.nu - 1 : a.nx, Edst = UP ? a.nx : a.nx + a.nu - 1;
    const double cpl = wd * so_element<NB>(v, Esrc);
    const bool has_u = k < a.NcT;
#pragma unroll
    for (int bi = 0; bi < NBLK; ++bi) {
        const bool live = q.isx[bi] || (q.isu[bi] && has_u);
        double r = UP ? sc[bi] * out[bi] : out[bi];
        r = live ? r : 0.0;
        if (q.e[bi] == Edst) r += cpl;
        t[bi] = r;
    }
}
template <int NB> struct SoStep { d4 S[SoCfg<NB>::NS]; double sc[NB / 16]; double wd; };
template <int NB>
__device__ __forceinline__ void so_step_load(const CoreArgs &a, const SoLane<NB> &q, int lane, int k, int nbr, SoStep<NB> &st) {
    // (middle stage of the S^-1-only format: omega straight from the metric vector.)  Loads only -- so_step_fix applies the
    // lane masks at consumption time.
    const int hi = max(k, nbr), lo = min(k, nbr);
    cgdouble *om = (cgdouble *)a.om;
#pragma unroll
    for (int bi = 0; bi < NB / 16; ++bi) st.sc[bi] = om[hi * a.nx + (q.isx[bi] ? q.e[bi] : 0)];
    st.wd = om[a.rdu + a.nu + min(lo, max(a.NcT - 2, 0)) * a.nu + a.nu - 1];      // (clamped: the row only exists for lo <= NcT-2)
    so_load<NB>(a.F + (size_t)k * a.fstage + FactorFmt<NB>::SOFF, lane, st.S);
}
template <int NB>
__device__ __forceinline__ void so_step_fix(const CoreArgs &a, const SoLane<NB> &q, int k, int nbr, const SoStep<NB> &st, double *sc, double &wd) {
#pragma unroll
    for (int bi = 0; bi < NB / 16; ++bi) sc[bi] = q.isx[bi] ? st.sc[bi] : 1.0;
    wd = (max(k, nbr) < a.NcT) ? st.wd : 0.0;       // the coupling row exists only between two stages that carry inputs
}
// ------------------------------------------------------------------------------------------------
// One half-chain of the formats that apply the off-diagonal blocks matrix-free, stages first+dir*i for i = ibegin .. nsteps:
//     SOLVE (forward sweep of the S^-1-only format, back substitution of the forward-matrix one):  v_k = S_k^-1 ( own_k - K_{k,nbr} v_nbr )
//     !SOLVE (back substitution of the S^-1-only format):                                   v_k = own_k - S_k^-1 K_{k,nbr} v_nbr
// own_k is what Tc holds for the stage, v_k replaces it; UP: the neighbour is the stage above (k-1, block from G), else below (G').
//     -K_{k,k-1} v = sc_k . (G v)   + cw_k v[nx+nu-1]          -K_{k,k+1} v = G' (sc_k . v) + cw_k v[nx]
// with (sc, cw) per element from the stage's table (FactorFmt) -- no row mask: S_k^-1 is stored with zero rows where the stage has
// no variable.  ibegin = 0 starts from a zero neighbour (v_first = S^-1 own_first), ibegin = 1 from v_first already in Tc.
//
// A sweeping wave issues in order, and at the large stage size it is alone on its SIMD: the loop is bound by the instructions it
// issues and by where it has to wait, not by memory (scripts/diag/stage_ubench.hip).  The stage is therefore laid out by hand:
// everything that does not depend on the running vector sits in the shadow of the stage's two dependent MFMA groups, pinned by
// scheduling fences (left alone, the scheduler puts loads, permutes and their waits in front of the first MFMA of the stage):
//     (1) G mat-vec issued            shadow A: selects that finish THIS stage's S^-1 (permutes issued a stage ago), LDS read of
//                                               the next own vector
//     (3) right-hand side, (4) S^-1 mat-vec issued
//                                     shadow B: cross-lane permutes for the NEXT stage's S^-1 (its packed half arrived a stage
//                                               ago), refill of this stage's slot (free once the MFMAs have read it)
//     (6) sum of the MFMA pairs, store
// (A gather of the fragment straight from the packed record -- 8-byte loads at per-lane offsets, no permutes -- was measured
// too: equal at 16 x 16, 12 % slower at 32 x 32, where twelve scattered loads per stage cost more in the address unit than
// twenty permutes in the LDS crossbar.)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pin_here(double &v) { asm volatile("" : "+v"(v)); }      // the value is computed before, and used after, this point
typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const d2 cgd2;
template <int NB> struct SoSlot { d4 S[SoCfg<NB>::NF]; d2 tab[NB / 16]; };
template <int NB> struct SoPerm { double s0[3], s1[3], tr[4]; };                 // permuted values on their way into a slot
struct SoLaneK { int pa[4]; bool m1, m2, m3; unsigned win, direct, tab; };      // per-lane constants: permute addresses, select masks, byte offsets
template <int NB, bool UP>
__device__ __forceinline__ SoLaneK so_lane_consts(int lane) {
    SoLaneK c;
    const int R = (lane >> 2) & 3, k = lane >> 4, i = lane & 3;
#pragma unroll
    for (int x = 0; x < 4; ++x) c.pa[x] = opaque_lane(4 * (16 * i + 4 * ((R + x) & 3) + k));
    c.m1 = R + 1 >= 4; c.m2 = R + 2 >= 4; c.m3 = R + 3 >= 4;
    c.win = (unsigned)opaque_lane(8 * (FactorFmt<NB>::SOFF + 40 * k + sym_cum(R) + i * (4 - R)));
    c.direct = (unsigned)opaque_lane(8 * (FactorFmt<NB>::SOFF + 328 + lane * 4));
    const int tabs = FactorFmt<NB>::SONLY && !UP ? 2 * NB : 0;                   // (S^-1-only: the table towards the stage below comes second)
    c.tab = (unsigned)opaque_lane(8 * (FactorFmt<NB>::SOFF + FactorFmt<NB>::SINV + tabs + 2 * vec_lane_offset(lane)));
    return c;
}
// packed record -> slot (loads only)
template <int NB>
__device__ __forceinline__ void so_slot_load(const char *Fk, const SoLaneK &c, SoSlot<NB> &s) {
    auto window = [&](unsigned extra) { const d4u w = *(cgd4u *)(Fk + extra + c.win); return d4{w[0], w[1], w[2], w[3]}; };
    if constexpr (NB == 16) {
        s.S[0] = window(0);
    } else { s.S[0] = window(0); s.S[3] = window(8 * 164); s.S[1] = *(cgd4 *)(Fk + c.direct); }
#pragma unroll
    for (int bi = 0; bi < NB / 16; ++bi) s.tab[bi] = *(cgd2 *)(Fk + c.tab + bi * 256);
}
// the same through buffer loads: rs = the instance's factor, soff = byte offset of the stage
template <int NB>
__device__ __forceinline__ void so_slot_load(__amdgpu_buffer_rsrc_t rs, unsigned soff, const SoLaneK &c, SoSlot<NB> &s) {
    if constexpr (NB == 16) {
        s.S[0] = buf_load_d4(rs, c.win, soff);
    } else { s.S[0] = buf_load_d4(rs, c.win, soff); s.S[3] = buf_load_d4(rs, c.win + 8 * 164, soff); s.S[1] = buf_load_d4(rs, c.direct, soff); }
#pragma unroll
    for (int bi = 0; bi < NB / 16; ++bi) {
        const bu4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, c.tab + bi * 256, soff, 0);
        s.tab[bi] = d2{bu_double(t[0], t[1]), bu_double(t[2], t[3])};
    }
}
// the cross-lane half of the expansion (sym_expand16 / frag_transpose16): issue ...
template <int NB>
__device__ __forceinline__ void so_expand_issue(const SoSlot<NB> &s, const SoLaneK &c, SoPerm<NB> &t) {
    {
        const d4 w = s.S[0];
        t.s0[0] = lane_permute(w[3], c.pa[1]); t.s0[1] = lane_permute(w[2], c.pa[2]); t.s0[2] = lane_permute(w[1], c.pa[3]);
    }
    if constexpr (NB == 32) {
        const d4 w = s.S[3], v = s.S[1];
        t.s1[0] = lane_permute(w[3], c.pa[1]); t.s1[1] = lane_permute(w[2], c.pa[2]); t.s1[2] = lane_permute(w[1], c.pa[3]);
        t.tr[0] = lane_permute(v[0], c.pa[0]); t.tr[1] = lane_permute(v[3], c.pa[1]); t.tr[2] = lane_permute(v[2], c.pa[2]); t.tr[3] = lane_permute(v[1], c.pa[3]);
    }
}
// ... and finish: the selects (pinned where they are written, or they sink to the first use)
template <int NB>
__device__ __forceinline__ void so_expand_finish(SoSlot<NB> &s, const SoLaneK &c, const SoPerm<NB> &t) {
    auto sel = [&](const d4 w, const double *p) {
        double e1 = c.m1 ? p[0] : w[1], e2 = c.m2 ? p[1] : w[2], e3 = c.m3 ? p[2] : w[3];
        pin_here(e1); pin_here(e2); pin_here(e3);
        return d4{w[0], e1, e2, e3};
    };
    s.S[0] = sel(s.S[0], t.s0);
    if constexpr (NB == 32) { s.S[3] = sel(s.S[3], t.s1); s.S[2] = d4{t.tr[0], t.tr[1], t.tr[2], t.tr[3]}; }
}
template <int NB, bool SOLVE, bool UP>
__device__ __forceinline__ void so_sweep(const CoreArgs &a, double *Tc, const int first, const int dir, const int ibegin, const int nsteps) {
    constexpr int NBLK = NB / 16, DEPTH = SoCfg<NB>::DEPTH;
    const int lane = opaque_lane(threadIdx.x & 63);
    double *tb = Tc + vec_lane_offset(lane);
    const bool writer = MPCQP_STORE_ALL ? true : vec_lane_writer(lane);
    const SoLaneK lc = so_lane_consts<NB, UP>(lane);
    d4 Gf[SoCfg<NB>::NF];
    frag_load<NB>(a.G + (UP ? 0 : NB * NB), lane, Gf);
    auto stage_of = [&](int i) { return first + dir * i; };
    auto clamp_i = [&](int i) { return i < nsteps ? i : nsteps; };       // (branch-free refills, see chain_sweep)
    const __amdgpu_buffer_rsrc_t rs = factor_rsrc(a.F);
    auto load = [&](int i, SoSlot<NB> &s) {
        if constexpr (MPCQP_BUFFER_LOADS && NB == 32) so_slot_load<NB>(rs, (unsigned)(stage_of(i) * a.fstage) * 8u, lc, s);
        else so_slot_load<NB>((const char *)(a.F + (size_t)stage_of(i) * a.fstage), lc, s);
    };
    if (nsteps < ibegin) return;
    double run[NBLK], own[NBLK];
#pragma unroll
    for (int bi = 0; bi < NBLK; ++bi) { run[bi] = ibegin ? tb[first * NB + 16 * bi] : 0.0; own[bi] = tb[stage_of(ibegin) * NB + 16 * bi]; }
    SoSlot<NB> ring[DEPTH];
    SoPerm<NB> pm;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        load(clamp_i(ibegin + d), ring[d]);
        __builtin_amdgcn_sched_barrier(0);
    }
    so_expand_issue<NB>(ring[0], lc, pm);                                 // the first stage has nothing to hide behind yet
    const int Esrc = UP ? a.nx + a.nu - 1 : a.nx;                         // the element of the neighbour that the Delta-u row couples
    auto stage_step = [&](int i, int inext, SoSlot<NB> &slot, SoSlot<NB> &nslot, bool valid) {
        const int k = stage_of(i);
        // (1) first dependent group: G (or G') times the neighbour's vector; beside it what does not wait for it
        double in[NBLK], p[NBLK], q[NBLK], base[NBLK];
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi) { in[bi] = UP ? run[bi] : slot.tab[bi][0] * run[bi]; p[bi] = 0.0; q[bi] = 0.0; }
#pragma unroll
        for (int bj = 0; bj < NBLK; ++bj) {
            const double r1 = rot_blocks<1>(in[bj]), r2 = rot_blocks<2>(in[bj]), r3 = rot_blocks<3>(in[bj]);
#pragma unroll
            for (int bi = 0; bi < NBLK; ++bi) {
                const d4 g = Gf[bi * NBLK + bj];
                p[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(g[0], in[bj], p[bi], 0, 0, 0);
                q[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(g[2], r2, q[bi], 0, 0, 0);
                p[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(g[1], r1, p[bi], 0, 0, 0);
                q[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(g[3], r3, q[bi], 0, 0, 0);
            }
        }
        const double nsrc = so_element<NB>(run, Esrc);
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi) { base[bi] = SOLVE ? fma(slot.tab[bi][1], nsrc, own[bi]) : slot.tab[bi][1] * nsrc; pin_here(base[bi]); }
        __builtin_amdgcn_sched_barrier(0);
        // shadow A: finish this stage's S^-1; the next stage's own vector
        so_expand_finish<NB>(slot, lc, pm);
        double own_next[NBLK];
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi) own_next[bi] = tb[stage_of(clamp_i(i + 1)) * NB + 16 * bi];
        __builtin_amdgcn_sched_barrier(0);
        // (3) what S_k^-1 multiplies
        double t[NBLK], xp[NBLK], xq[NBLK];
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi) { t[bi] = UP ? fma(slot.tab[bi][0], p[bi] + q[bi], base[bi]) : (p[bi] + q[bi]) + base[bi]; xp[bi] = 0.0; xq[bi] = 0.0; }
        // (4) second dependent group
#pragma unroll
        for (int bj = 0; bj < NBLK; ++bj) {
            const double r1 = rot_blocks<1>(t[bj]), r2 = rot_blocks<2>(t[bj]), r3 = rot_blocks<3>(t[bj]);
#pragma unroll
            for (int bi = 0; bi < NBLK; ++bi) {
                const d4 sf = slot.S[bi * NBLK + bj];
                xp[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(sf[0], t[bj], xp[bi], 0, 0, 0);
                xq[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(sf[2], r2, xq[bi], 0, 0, 0);
                xp[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(sf[1], r1, xp[bi], 0, 0, 0);
                xq[bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(sf[3], r3, xq[bi], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // shadow B: permutes for the next stage's S^-1 (loaded a stage ago), refill of this slot
        so_expand_issue<NB>(nslot, lc, pm);
        load(inext, slot);
        __builtin_amdgcn_sched_barrier(0);
        // (6)
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi) { run[bi] = SOLVE ? xp[bi] + xq[bi] : (xp[bi] + xq[bi]) + own[bi]; own[bi] = own_next[bi]; }
        if (writer && valid) {
#pragma unroll
            for (int bi = 0; bi < NBLK; ++bi) tb[k * NB + bi * 16] = run[bi];
        }
    };
    for (int i0 = ibegin; i0 <= nsteps; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) stage_step(clamp_i(i0 + d), clamp_i(i0 + d + DEPTH), ring[d], ring[(d + 1) % DEPTH], i0 + d <= nsteps);
    }
}

template <int NB>
__device__ __forceinline__ void kkt_core_so(const CoreArgs &a, double *Tc) {
    constexpr int NBLK = NB / 16, NF = SoCfg<NB>::NF;
    const int N = a.N, mid = N / 2, wv = logical_wave(), lane = opaque_lane(threadIdx.x & 63);
    TICK_START
    if (wv == 0) so_sweep<NB, true, true>(a, Tc, 0, +1, 0, mid - 1);                     // w_0 .. w_{mid-1}
    else if (wv == 1) so_sweep<NB, true, false>(a, Tc, N - 1, -1, 0, N - 2 - mid);       // w_{N-1} .. w_{mid+1}
    __syncthreads();
    TICK(1)
    if (wv == 0) {                                                                       // the middle stage sees both halves
        const SoLane<NB> q = so_lane<NB>(a, lane);
        double *tb = Tc + vec_lane_offset(lane);
        SoStep<NB> su, sd;
        so_step_load<NB>(a, q, lane, mid, mid - 1, su);
        so_step_load<NB>(a, q, lane, mid, mid + 1, sd);
        d4 Gf[NF], Sf[NF];
        double vu[NBLK], vd[NBLK], own[NBLK], tu[NBLK], td[NBLK], out[NBLK];
        vec_load<NB>(tb, mid - 1, vu); vec_load<NB>(tb, mid + 1, vd); vec_load<NB>(tb, mid, own);
        frag_load<NB>(a.G, lane, Gf);
        double scu[NBLK], scd[NBLK], wdu, wdd;
        so_step_fix<NB>(a, q, mid, mid - 1, su, scu, wdu);
        so_step_fix<NB>(a, q, mid, mid + 1, sd, scd, wdd);
        so_offdiag<NB, true>(a, q, Gf, mid, vu, scu, wdu, tu);
        frag_load<NB>(a.G + NB * NB, lane, Gf);
        so_offdiag<NB, false>(a, q, Gf, mid, vd, scd, wdd, td);
        so_expand<NB>(su.S, lane, Sf);
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi) { tu[bi] += own[bi] + td[bi]; out[bi] = 0.0; }
        frag_matvec<NB>(Sf, tu, out);
        vec_store<NB>(tb, mid, out, vec_lane_writer(lane));
    }
    __syncthreads();
    TICK(2)
    if (wv == 0) so_sweep<NB, false, false>(a, Tc, mid, -1, 1, mid);                     // x_{mid-1} .. x_0       (neighbour below)
    else if (wv == 1) so_sweep<NB, false, true>(a, Tc, mid, +1, 1, N - 1 - mid);         // x_{mid+1} .. x_{N-1}   (neighbour above)
    __syncthreads();
    TICK(3)
}

// Forward-matrix format: forward elimination from the forward matrices (chain_sweep), the middle stage, then back substitution
//     x_k = S_k^-1 ( yh_k - K_{k,nbr} x_nbr )
// outwards from the middle with the off-diagonal blocks applied matrix-free -- S^-1 is read once, the forward matrices once.
template <int NB>
__device__ __forceinline__ void kkt_core_fwd(const CoreArgs &a, double *Tc) {
    constexpr int NBLK = NB / 16, NF = SoCfg<NB>::NF;
    const int N = a.N, fstage = a.fstage, mid = N / 2, wv = logical_wave(), lane = opaque_lane(threadIdx.x & 63);
    const double *F = a.F;
    TICK_START
    if (wv == 0) chain_sweep<NB>(0, +1, mid - 1, fstage, F, Tc);                          // yh_1 .. yh_{mid-1}
    else if (wv == 1) chain_sweep<NB>(N - 1, -1, N - 2 - mid, fstage, F, Tc);             // yh_{N-2} .. yh_{mid+1}
    __syncthreads();
    TICK(1)
    if (wv == 0) {                                   // yh_mid = b_mid - Mh_mid yh_{mid-1} - Mt_mid yh_{mid+1};  x_mid = S_mid^-1 yh_mid
        double *tb = Tc + vec_lane_offset(lane);
        d4 A0[NF], A2[NF], Am[SoCfg<NB>::NS], Sf[NF];
        frag_load<NB>(F + (size_t)mid * fstage, lane, A0);
        frag_load<NB>(F, lane, A2);                  // the middle's second forward matrix (kept in stage 0's slot)
        so_load<NB>(F + (size_t)mid * fstage + FactorFmt<NB>::SOFF, lane, Am);
        double up[NBLK], dn[NBLK], acc[NBLK], out[NBLK];
        vec_load<NB>(tb, mid, acc); vec_load<NB>(tb, mid - 1, up); vec_load<NB>(tb, mid + 1, dn);
        frag_matvec<NB>(A0, up, acc);
        frag_matvec<NB>(A2, dn, acc);
        so_expand<NB>(Am, lane, Sf);
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi) out[bi] = 0.0;
        frag_matvec<NB>(Sf, acc, out);
        vec_store<NB>(tb, mid, out, vec_lane_writer(lane));
    }
    __syncthreads();
    TICK(2)
    if (wv == 0) so_sweep<NB, true, false>(a, Tc, mid, -1, 1, mid);                        // x_{mid-1} .. x_0       (neighbour below)
    else if (wv == 1) so_sweep<NB, true, true>(a, Tc, mid, +1, 1, N - 1 - mid);            // x_{mid+1} .. x_{N-1}   (neighbour above)
    __syncthreads();
    TICK(3)
}

// Tc <- K_xu^-1 Tc (eps already eliminated).  All threads call; barriers inside.  Waves 0 and 1 sweep the two
// half-chains of the twisted factorization concurrently.  Tc must be seen by the compiler as an LDS pointer
// (a pointer laundered through an integer becomes FLAT: flat LDS accesses count on vmcnt AND lgkmcnt and force a
// full s_waitcnt vmcnt(0) -- draining the factor prefetch -- before every stage).
__device__ __forceinline__ void kkt_core_group(const CoreArgs &, double *);      // mpcqp_group.h: several small stages per 16 x 16 block
// (GROUPABLE = false: an instantiation that can never meet grouped stages -- compile-time nx + nu = 16 -- does not carry that path)
template <int NB, bool GROUPABLE = (NB == 16)>
__device__ __forceinline__ void kkt_core(const CoreArgs &a, double *Tc) {
#ifndef MPCQP_ABL_NOCHAIN
    if constexpr (NB == 16 && GROUPABLE) { if (a.grp > 1) { kkt_core_group(a, Tc); return; } }
    if constexpr (FactorFmt<NB>::SONLY) kkt_core_so<NB>(a, Tc);
    else kkt_core_fwd<NB>(a, Tc);                 // (each of them ends with a barrier)
#else
    __syncthreads();
#endif
}
template <> __device__ __forceinline__ void kkt_core<64, false>(const CoreArgs &, double *);      // mpcqp_wide.h (stages wider than 32)
template <> __device__ __forceinline__ void kkt_core<128, false>(const CoreArgs &, double *);     // mpcqp_huge.h (stages wider than 64)
