// mpcqp_latw.h -- part of libmpcqp_hip (included by mpcqp.hip and by mpcqp_w8.hip, one translation unit each).
// The latency round of mpcqp_lat.h (block cyclic reduction, factor and iterate resident on the compute unit) for ANY number of waves per
// workgroup -- built for EIGHT (mpcqp_w8.hip: NT = 512, two waves per SIMD), also instantiated for four -- with a DENSE TOP (MODE_BCRT + N).
//
// What the four-wave kernel of mpcqp_lat.h spends an iteration of one (12,4,30) instance on (cycles, alone on its compute unit): owner
// passes 1 370 + 820, level 0 forward 2 290 / back 1 570, level 1 forward 1 240 / back 1 060, levels 2 .. 4 on ONE wave 2 700.  None of it
// is issue-bound: a wave's 11 mat-vecs of level 0 are 46 MFMAs = 780 cycles of matrix-pipe issue and take 2 100 -- in-order waves, one per
// SIMD, waiting on dependent MFMAs (44 cycles), on LDS round trips and on double-precision vector instructions (32 cycles dependent).  Two
// changes, both about that:
//   * eight waves: every level step and both owner passes have work for all of them (half a group of four stages' owner items and half the
//     mat-vecs per wave), and the second wave of a SIMD issues into the first one's stalls.  A wave then has 256 registers instead of 512, so
//     it holds its <= 15 level fragments (120 registers) and no more:
//   * the levels above 1 are replaced by the explicit inverse of what two levels leave (BcrFmt, factor_bcr): nt = N / 4 stages (7 of 31), nt
//     block rows of nt mat-vecs, one row per wave, ONE barrier instead of a five-deep chain on one wave.  Its nt^2 fragments (49: 98 KB) are
//     what the registers no longer hold: they sit in LDS for the whole launch (with the constant fragments G, G'; a refactorization says when
//     they are stale) -- the latency kernels use 44 of the compute unit's 160 KB otherwise.  Since the second half of round 5 the top is a
//     VECTOR-ALU mat-vec over that LDS copy (LATW_TOP_VALU below: 2 043 -> 1 214 cycles of an iteration's 8 138); the matrix-core form
//     (fragments as [block][half][lane][2 doubles], a lane's 32 bytes two conflict-free 16-byte reads) remains as a build switch.
// Barriers per iteration: seven as before (G'W + right-hand side | level 0 | level 1 | top | level 1 back | level 0 back | G v + row updates) where the waves share
// groups of stages; FIVE where every wave owns exactly one group of four stages (31 stages on eight waves: LATW_FUSE0 below) -- a wave then runs the level-0 tasks of
// its own stages straight behind its right-hand side, and both back substitutions of its own stages in one phase.
#pragma once

// ---- static schedule of levels 0 and 1 over NWAVES waves (task kinds as in mpcqp_lat.h: 0 kept stage forward, 1 eliminated stage forward, 2 back substitution)
// kept stages (two mat-vecs each) are dealt from wave 0 up, eliminated ones (one mat-vec) from the last wave down
constexpr int latw_owner(int /*N*/, int /*L*/, int kind, int t) { return kind == 1 ? NWAVES - 1 - t % NWAVES : t % NWAVES; }
// LATW_FUSE0: level 0 forward WITHOUT the barrier behind the right-hand side, where every wave owns exactly one group of four stages (31 stages on
// eight waves).  Wave w then runs the level-0 tasks of ITS stages 4w .. 4w+3 -- D^-1 of 4w and 4w+2, the kept stage 4w+1, the kept stage 4w+3 from
// its left neighbour 4w+2 -- straight from the right-hand side it has just written (a wave reads its own LDS writes in order), plus the one
// contribution that crosses into the previous group: the right neighbour 4w of the kept stage 4w-1, left in a second vector (x2) that level 1
// forward adds when it starts that stage's chain.  Six fragments per wave as before; one barrier and the wait in front of it less per iteration.
// LATW_FUSE0 >= 2: the same idea on the way back -- level 1 back and level 0 back of the wave's own stages in ONE phase: x(4w+1) from the top's solution
// at 4w-1 and 4w+3 (read where the top left it), then x(4w) and x(4w+2) from those three -- nothing another wave computes in the same phase.
// Five barriers per iteration: right-hand side + level 0 | level 1 | top | back | owner update.  Measured (one (12,4,30) instance alone, cycles per iteration; 128 / 256
// instances): unfused 7 407, 690 k / 1.15 M solves/s; level 0 forward fused 6 966, 724 k / 1.21 M; back substitution fused too 6 372, 773 k / 1.29 M.  Level 1 forward
// in the same phase as well (its tasks need only b'(4w+1), the top then adds two partial inputs as it reads them; four barriers): 6 938, 724 k / 1.19 M -- nine
// dependent mat-vecs on one wave are longer than six and three with a barrier between them, and the extra live values spill in the owner passes; not kept.
#ifndef LATW_FUSE0
#define LATW_FUSE0 2
#endif
constexpr bool latw_fused(int N) { return LATW_FUSE0 && (N + 3) / 4 == NWAVES; }
constexpr bool latw_fused_back(int N) { return LATW_FUSE0 >= 2 && latw_fused(N); }
constexpr int LATW_F0 = 6;          // fragment slots of the fused level-0 forward: D^-1(4w), D^-1(4w+2), kept 4w+1 (left, right), kept 4w+3 (left), kept 4w-1 (right)
constexpr int LATW_FB = 6;          // ... of the fused back substitution: x(4w+1) (left, right), x(4w) (left, right), x(4w+2) (left, right)
// slot of a task's first fragment in its wave's register array: level 0 forward, level 1 forward, level 1 back, level 0 back
constexpr int latw_slot(int N, int W, int Lq, int kq, int tq) {
    int s = latw_fused(N) ? LATW_F0 : 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int li = 0; li < 2; ++li) {
            const int L = pass == 0 ? li : 1 - li;
            for (int kind = (pass == 0 ? 0 : 2); kind < (pass == 0 ? 2 : 3); ++kind)
                for (int t = 0; t < lat_count(N, L, kind); ++t) {
                    if (latw_fused(N) && L == 0 && kind < 2) continue;
                    if (latw_fused_back(N) && kind == 2) continue;
                    if (L == Lq && kind == kq && t == tq) return s;
                    if (latw_owner(N, L, kind, t) == W) s += lat_nfr(N, L, kind, t);
                }
        }
    return s;
}
constexpr int latw_back_base(int N, int W) { return latw_slot(N, W, -1, -1, -1); }      // (fused back substitution: its six slots behind everything scheduled)
constexpr int latw_slots(int N, int W) { return latw_slot(N, W, -1, -1, -1) + (latw_fused_back(N) ? LATW_FB : 0); }
constexpr int latw_max_slots(int N) { int m = 0; for (int w = 0; w < NWAVES; ++w) m = latw_slots(N, w) > m ? latw_slots(N, w) : m; return m; }
constexpr int latw_top_stage(int r) { return 4 * (r + 1) - 1; }

// The top on the VECTOR ALU (LATW_TOP_VALU; 0 = on the matrix cores, latw_top, as until the middle of round 5): a mat-vec uses one of the four
// B columns of v_mfma_f64_4x4x4, so the 28 MFMAs a wave issues for its block row carry 64 useful products each at 17-20 cycles of the matrix
// pipe; the same 64 products are ONE v_fma_f64 at 6.5 cycles when enough of them are independent (scripts/diag/mfma_rate.hip).  The inverse then
// sits in LDS in row-part order instead of fragment order, the reduced right-hand sides of the top stages in a compact vector Xt behind it:
//   1: lane (i = lane & 15, p = lane >> 4) of the wave that owns block row r: row 16 r + i times columns [4 nt p, 4 nt (p + 1)) -- 2 nt 16-byte reads
//      of the matrix, 2 nt 16-byte broadcast reads of Xt, 4 nt FMAs on 7 independent accumulators, two cross-lane steps (p);
//   2: lane (q = lane >> 3, p = lane & 7): rows 16 r + 2 q, 2 q + 1 times columns [2 nt p, 2 nt (p + 1)) -- half the reads of Xt, three DPP steps.
// Measured (one (12,4,30) instance alone, cycles of the top phase / of the iteration; 128 / 256 instances, driver's flags):
//   matrix cores 2 043 / 8 138, 628 k / 1.04 M solves/s;  mode 1 (groups of 2, pipelined) 1 464 / 7 545, 654 k / 1.08 M;  mode 2 (the same) 1 214 / 7 310,
//   682 k / 1.13 M;  larger groups (3 or 4 column pairs in flight twice) spill in the owner passes and lose more there than they gain here.
// (LATW_TOP_VALU and the order of the inverse: mpcqp_bcr.h)
#define LATW_TOP_LDS(N) ((BcrFmt::top_count(N) * BcrFmt::top_count(N) + 2) * BcrFmt::NN + 32 * BcrFmt::top_count(N))      /* LDS doubles of the top inverse and, behind it, of the fragments G and G', of Xt and of x2 */
#if NT == 512
#define LATW_DISPATCH(wv, CALL) switch (wv) { \
    case 0: { constexpr int W = 0; CALL; } break; case 1: { constexpr int W = 1; CALL; } break; \
    case 2: { constexpr int W = 2; CALL; } break; case 3: { constexpr int W = 3; CALL; } break; \
    case 4: { constexpr int W = 4; CALL; } break; case 5: { constexpr int W = 5; CALL; } break; \
    case 6: { constexpr int W = 6; CALL; } break; default: { constexpr int W = 7; CALL; } break; }
#else
#define LATW_DISPATCH(wv, CALL) LAT_DISPATCH(wv, CALL)
#endif

template <int N, int W>
__device__ __forceinline__ void latw_load(const double *F, d4 *fr) {
    const int lane = threadIdx.x & 63;
    if constexpr (latw_fused(N)) {
        constexpr int s0 = 4 * W;
        if constexpr (s0 < N) { fr[0] = bcr_frag(F, s0, BcrFmt::ODINV, lane); fr[2] = bcr_frag(F, s0, BcrFmt::OLBRT, lane); }
        if constexpr (s0 + 2 < N) { fr[1] = bcr_frag(F, s0 + 2, BcrFmt::ODINV, lane); fr[3] = bcr_frag(F, s0 + 2, BcrFmt::OLBLT, lane); }
        if constexpr (s0 + 3 < N) fr[4] = bcr_frag(F, s0 + 2, BcrFmt::OLBRT, lane);
        if constexpr (W > 0 && s0 < N) fr[5] = bcr_frag(F, s0, BcrFmt::OLBLT, lane);
    }
    if constexpr (latw_fused_back(N)) {
        constexpr int bb = latw_back_base(N, W), e1 = 4 * W + 1, ea = 4 * W, eb = 4 * W + 2;
        if constexpr (e1 < N) { if constexpr (e1 - 2 >= 0) fr[bb] = bcr_frag(F, e1, BcrFmt::OLBL, lane); if constexpr (e1 + 2 < N) fr[bb + 1] = bcr_frag(F, e1, BcrFmt::OLBR, lane); }
        if constexpr (ea < N) { if constexpr (ea - 1 >= 0) fr[bb + 2] = bcr_frag(F, ea, BcrFmt::OLBL, lane); if constexpr (ea + 1 < N) fr[bb + 3] = bcr_frag(F, ea, BcrFmt::OLBR, lane); }
        if constexpr (eb < N) { fr[bb + 4] = bcr_frag(F, eb, BcrFmt::OLBL, lane); if constexpr (eb + 1 < N) fr[bb + 5] = bcr_frag(F, eb, BcrFmt::OLBR, lane); }
    }
    static_for<0, 2>([&](auto lc) {
        constexpr int L = decltype(lc)::value, h = 1 << L;
        static_for<0, lat_count(N, L, 0)>([&](auto tc) {
            constexpr int t = decltype(tc)::value, i = lat_stage(L, 0, t);
            if constexpr (!(latw_fused(N) && L == 0) && latw_owner(N, L, 0, t) == W) {
                constexpr int s = latw_slot(N, W, L, 0, t);
                fr[s] = bcr_frag(F, i - h, BcrFmt::OLBRT, lane);
                if constexpr (i + h < N) fr[s + 1] = bcr_frag(F, i + h, BcrFmt::OLBLT, lane);
            }
        });
        static_for<0, lat_count(N, L, 1)>([&](auto tc) {
            constexpr int t = decltype(tc)::value, e = lat_stage(L, 1, t);
            if constexpr (!(latw_fused(N) && L == 0) && latw_owner(N, L, 1, t) == W) { constexpr int s1 = latw_slot(N, W, L, 1, t); fr[s1] = bcr_frag(F, e, BcrFmt::ODINV, lane); }
            if constexpr (!latw_fused_back(N) && latw_owner(N, L, 2, t) == W) {
                constexpr int s = latw_slot(N, W, L, 2, t);
                if constexpr (e - h >= 0) fr[s] = bcr_frag(F, e, BcrFmt::OLBL, lane);
                constexpr int s2 = s + (e - h >= 0 ? 1 : 0);
                if constexpr (e + h < N) fr[s2] = bcr_frag(F, e, BcrFmt::OLBR, lane);
            }
        });
    });
}

// LDS vectors of the round (stage-major, stride 16; mpcqp_lat.h): tb right-hand side / solution, cb c_e of the reduction -- and, at the top stages'
// slots, the top's solution until level 1 back has copied it into tb; each seen through the lane bases of the four block rotations.
struct LatwVecs { double *tb, *cb; const double *t1, *t2, *t3, *c1, *c2, *c3; double *xt, *x2; };      // (xt: LATW_TOP_VALU, the top's compact input, per-lane base as tb; x2: LATW_FUSE0, the top stages' contributions from the next group)
#ifndef LATW_IN_DPP
#define LATW_IN_DPP 0              // 1: one LDS read per input vector and three DPP block rotations instead of four reads (measured: see LAB_NOTES.md)
#endif
template <bool FROMC>
__device__ __forceinline__ void latw_mv_lds(const d4 a, const LatwVecs &v, int off, double &p, double &q) {
#if LATW_IN_DPP
    lat_mv(a, FROMC ? v.cb[off] : v.tb[off], p, q);
    return;
#endif
    const double i0 = FROMC ? v.cb[off] : v.tb[off], i1 = FROMC ? v.c1[off] : v.t1[off], i2 = FROMC ? v.c2[off] : v.t2[off], i3 = FROMC ? v.c3[off] : v.t3[off];
    p = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], i0, p, 0, 0, 0);
    q = __builtin_amdgcn_mfma_f64_4x4x4f64(a[2], i2, q, 0, 0, 0);
    p = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], i1, p, 0, 0, 0);
    q = __builtin_amdgcn_mfma_f64_4x4x4f64(a[3], i3, q, 0, 0, 0);
}

template <int N, int W, int L>
__device__ __forceinline__ void latw_fwd(const d4 *fr, const LatwVecs &v) {
    constexpr int h = 1 << L;
    static_for<0, lat_count(N, L, 0)>([&](auto tc) {
        constexpr int t = decltype(tc)::value, i = lat_stage(L, 0, t);
        if constexpr (latw_owner(N, L, 0, t) == W) {
            constexpr int s = latw_slot(N, W, L, 0, t);
            double p = v.tb[i * 16], q = 0.0;                 // (the stage's own right-hand side starts the chain)
            if constexpr (latw_fused(N) && L == 1) p += v.x2[((i + 1) / 4 - 1) * 16];      // (its level-0 contribution from the next group's first stage)
            latw_mv_lds<false>(fr[s], v, (i - h) * 16, p, q);
            if constexpr (i + h < N) latw_mv_lds<false>(fr[s + 1], v, (i + h) * 16, p, q);
            // (level 1 keeps exactly the top stages 4 (r + 1) - 1: with the vector-ALU top their reduced right-hand side goes to the compact Xt --
            //  nobody reads tb there before level 1 back has put the top's solution in its place)
            if constexpr (LATW_TOP_VALU && L == 1) v.xt[((i + 1) / 4 - 1) * 16] = p + q;
            else v.tb[i * 16] = p + q;
        }
    });
    static_for<0, lat_count(N, L, 1)>([&](auto tc) {
        constexpr int t = decltype(tc)::value, e = lat_stage(L, 1, t);
        if constexpr (latw_owner(N, L, 1, t) == W) {
            constexpr int s1 = latw_slot(N, W, L, 1, t);
            double p = 0.0, q = 0.0;
            latw_mv_lds<false>(fr[s1], v, e * 16, p, q);
            v.cb[e * 16] = p + q;
        }
    });
}
// level 0 forward of wave W's own four stages (LATW_FUSE0), called right behind the right-hand side without a barrier
template <int N, int W>
__device__ __forceinline__ void latw_fwd0_own(const d4 *fr, const LatwVecs &v) {
    constexpr int s0 = 4 * W;
    if constexpr (s0 < N) { double p = 0.0, q = 0.0; latw_mv_lds<false>(fr[0], v, s0 * 16, p, q); v.cb[s0 * 16] = p + q; }
    if constexpr (s0 + 2 < N) { double p = 0.0, q = 0.0; latw_mv_lds<false>(fr[1], v, (s0 + 2) * 16, p, q); v.cb[(s0 + 2) * 16] = p + q; }
    if constexpr (s0 + 1 < N) {
        double p = v.tb[(s0 + 1) * 16], q = 0.0;
        latw_mv_lds<false>(fr[2], v, s0 * 16, p, q);
        if constexpr (s0 + 2 < N) latw_mv_lds<false>(fr[3], v, (s0 + 2) * 16, p, q);
        v.tb[(s0 + 1) * 16] = p + q;
    }
    if constexpr (s0 + 3 < N) { double p = v.tb[(s0 + 3) * 16], q = 0.0; latw_mv_lds<false>(fr[4], v, (s0 + 2) * 16, p, q); v.tb[(s0 + 3) * 16] = p + q; }
    if constexpr (W > 0 && s0 < N) { double p = 0.0, q = 0.0; latw_mv_lds<false>(fr[5], v, s0 * 16, p, q); v.x2[(W - 1) * 16] = p + q; }
}
// back substitution of level L; TOPIN: the neighbours are top stages whose solution still sits in cb (level 1)
template <int N, int W, int L, bool TOPIN>
__device__ __forceinline__ void latw_bwd(const d4 *fr, const LatwVecs &v) {
    constexpr int h = 1 << L;
    static_for<0, lat_count(N, L, 2)>([&](auto tc) {
        constexpr int t = decltype(tc)::value, e = lat_stage(L, 2, t);
        if constexpr (latw_owner(N, L, 2, t) == W) {
            constexpr int s = latw_slot(N, W, L, 2, t), s2 = s + (e - h >= 0 ? 1 : 0);
            double p = v.cb[e * 16], q = 0.0;
            if constexpr (e - h >= 0) latw_mv_lds<TOPIN>(fr[s], v, (e - h) * 16, p, q);
            if constexpr (e + h < N) latw_mv_lds<TOPIN>(fr[s2], v, (e + h) * 16, p, q);
            v.tb[e * 16] = p + q;
        }
    });
}

template <int N, int W>
__device__ __forceinline__ void latw_bwd1(const d4 *fr, const LatwVecs &v) {
    static_for<0, BcrFmt::top_count(N)>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if constexpr (r % NWAVES == W) v.tb[latw_top_stage(r) * 16] = v.cb[latw_top_stage(r) * 16];
    });
    latw_bwd<N, W, 1, true>(fr, v);
}

// level 1 back and level 0 back of wave W's own stages in one phase (LATW_FUSE0 >= 2); the top's solution is read from cb where the neighbour is a top stage
template <int N, int W>
__device__ __forceinline__ void latw_bwd_own(const d4 *fr, const LatwVecs &v) {
    constexpr int bb = latw_back_base(N, W), e1 = 4 * W + 1, ea = 4 * W, eb = 4 * W + 2;
    if constexpr (W < BcrFmt::top_count(N)) v.tb[latw_top_stage(W) * 16] = v.cb[latw_top_stage(W) * 16];      // (the top's row W into tb, for the owner passes)
    if constexpr (e1 < N) {
        double p = v.cb[e1 * 16], q = 0.0;
        if constexpr (e1 - 2 >= 0) latw_mv_lds<true>(fr[bb], v, (e1 - 2) * 16, p, q);
        if constexpr (e1 + 2 < N) latw_mv_lds<true>(fr[bb + 1], v, (e1 + 2) * 16, p, q);
        v.tb[e1 * 16] = p + q;
    }
    if constexpr (ea < N) {
        double p = v.cb[ea * 16], q = 0.0;
        if constexpr (ea - 1 >= 0) latw_mv_lds<true>(fr[bb + 2], v, (ea - 1) * 16, p, q);
        if constexpr (ea + 1 < N) latw_mv_lds<false>(fr[bb + 3], v, (ea + 1) * 16, p, q);
        v.tb[ea * 16] = p + q;
    }
    if constexpr (eb < N) {
        double p = v.cb[eb * 16], q = 0.0;
        latw_mv_lds<false>(fr[bb + 4], v, (eb - 1) * 16, p, q);
        if constexpr (eb + 1 < N) latw_mv_lds<true>(fr[bb + 5], v, (eb + 1) * 16, p, q);
        v.tb[eb * 16] = p + q;
    }
}

// one fragment of the top inverse from its LDS copy: [block][half][lane][2]
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d4 latw_top_frag(const double *TopL, int blk, int lane) {
    const d2 lo = *(const d2 *)(TopL + ((blk * 2 + 0) * 64 + lane) * 2), hi = *(const d2 *)(TopL + ((blk * 2 + 1) * 64 + lane) * 2);
    return d4{lo[0], lo[1], hi[0], hi[1]};
}
// the top: block row r of the inverse times the reduced right-hand sides of the nt top stages (tb), into cb at stage i_r.
// Measured on the way (cycles of this phase for one (12,4,30) instance alone, 8 waves; scripts/lat_phase.py with a -DMPCQP_RUN_TIMING build):
//   * left to itself the compiler requests a row's fragments one or two at a time, each right in front of the MFMAs that need it (it has ~ 30
//     free registers), and the wave sits through an LDS round trip per block: 2 115 - 2 340;
//   * all fragments of the row up front, inputs read once and rotated by DPP moves: 2 320 -- the 42 moves alone cost 600 (without them 1 810),
//     the MFMAs alone 1 120 (28 per wave, two waves per SIMD: 20 cycles each, the matrix pipe's rate), the fragment reads alone 980;
//   * independent accumulators per column instead of two pairs of chains: 2 380; the chains of two columns interleaved MFMA by MFMA: 2 200;
//   * inputs as four 8-byte LDS reads through the rotated lane bases, the row in groups of LATW_TOP_HALF columns whose fragments and inputs
//     are requested together and whose chains are interleaved: 2 columns 1 975 (kept), 4: 2 206, all 7: 2 270 and spills in other phases.
// The phase is at twice its matrix-pipe time; what is left is LDS delivery (98 KB of fragments per iteration) not overlapping with it.
#ifndef LATW_TOP_HALF
#define LATW_TOP_HALF 2
#endif
template <int N, int W>
__device__ __forceinline__ void latw_top(const double *TopL, const LatwVecs &v, int lane) {
    constexpr int NTOP = BcrFmt::top_count(N), H = LATW_TOP_HALF;
    static_for<0, NTOP>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if constexpr (r % NWAVES == W) {
            double p0 = 0.0, q0 = 0.0, p1 = 0.0, q1 = 0.0;
            static_for<0, (NTOP + H - 1) / H>([&](auto hc) {
                constexpr int c0 = decltype(hc)::value * H, c1 = c0 + H < NTOP ? c0 + H : NTOP, n = c1 - c0;
                d4 a[H]; double x[H][4];
                static_for<0, n>([&](auto cc_) {
                    constexpr int j = decltype(cc_)::value, off = latw_top_stage(c0 + j) * 16;
                    a[j] = latw_top_frag(TopL, r * NTOP + c0 + j, lane);
                    x[j][0] = v.tb[off]; x[j][1] = v.t1[off]; x[j][2] = v.t2[off]; x[j][3] = v.t3[off];
                });
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, (n + 1) / 2>([&](auto pc) {
                    constexpr int j = 2 * decltype(pc)::value;
                    if constexpr (j + 1 < n) {
                        p0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j][0], x[j][0], p0, 0, 0, 0); p1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j + 1][0], x[j + 1][0], p1, 0, 0, 0);
                        q0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j][2], x[j][2], q0, 0, 0, 0); q1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j + 1][2], x[j + 1][2], q1, 0, 0, 0);
                        p0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j][1], x[j][1], p0, 0, 0, 0); p1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j + 1][1], x[j + 1][1], p1, 0, 0, 0);
                        q0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j][3], x[j][3], q0, 0, 0, 0); q1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j + 1][3], x[j + 1][3], q1, 0, 0, 0);
                    } else {
                        p0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j][0], x[j][0], p0, 0, 0, 0); q0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j][2], x[j][2], q0, 0, 0, 0);
                        p0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j][1], x[j][1], p0, 0, 0, 0); q0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j][3], x[j][3], q0, 0, 0, 0);
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            v.cb[latw_top_stage(r) * 16] = (p0 + q0) + (p1 + q1);
        }
    });
}

// ---- the top on the vector ALU (LATW_TOP_VALU; order of the inverse in LDS and in memory: bcr_topv_rc, mpcqp_bcr.h) ----------------------------------------
template <int CTRL>
__device__ __forceinline__ double latw_dpp(double x) {
    const long long xi = __builtin_bit_cast(long long, x);
    const int lo = DPP_MOVE((int)xi, CTRL), hi = DPP_MOVE((int)(xi >> 32), CTRL);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
// Left to itself the compiler requests one or two 16-byte reads at a time and waits for each (an LDS round trip per pair of FMAs: it has ~ 30
// free registers here), so the reads are issued in groups of LATW_TOPV_GRP column pairs behind scheduling fences; LATW_TOPV_PIPE: the next
// group's reads are issued before the current group's FMAs (two groups of registers in flight).  (Requesting a wave's first group of matrix
// pairs on the other side of the barrier, at the start of level 1 forward -- they depend on nothing the iteration computes -- took 80 cycles off
// this phase and, through 16 more live registers, added 300 to the owner passes: 7 572 against 7 352 cycles per iteration; not kept.)
#ifndef LATW_TOPV_GRP
#define LATW_TOPV_GRP 2
#endif
#ifndef LATW_TOPV_PIPE
#define LATW_TOPV_PIPE 1
#endif
template <int N, int W>
__device__ __forceinline__ void latw_top_valu(const double *TopL, const double *Xt, double *Cc, int lane) {
    constexpr int NTOP = BcrFmt::top_count(N), H = LATW_TOPV_GRP;
    static_for<0, NTOP>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if constexpr (r % NWAVES == W) {
            const d2 *tm = (const d2 *)TopL + (size_t)(r * 2 * NTOP) * 64 + lane;
#if LATW_TOP_VALU == 2
            constexpr int KQ = NTOP, NA = NTOP < 4 ? NTOP : 4, NG = (KQ + H - 1) / H;
            const d2 *xv = (const d2 *)(Xt + 2 * NTOP * (lane & 7));
            double a0[NA], a1[NA];
            d2 bx[2][H], b0[2][H], b1[2][H];
            auto load = [&](auto gc) {
                constexpr int g = decltype(gc)::value, k0 = g * H, n = (k0 + H < KQ ? k0 + H : KQ) - k0;
                static_for<0, n>([&](auto jc) { constexpr int j = decltype(jc)::value, k = k0 + j; bx[g & 1][j] = xv[k]; b0[g & 1][j] = tm[k * 64]; b1[g & 1][j] = tm[(NTOP + k) * 64]; });
            };
            auto compute = [&](auto gc) {
                constexpr int g = decltype(gc)::value, k0 = g * H, n = (k0 + H < KQ ? k0 + H : KQ) - k0;
                static_for<0, n>([&](auto jc) {
                    constexpr int j = decltype(jc)::value, k = k0 + j, q = k % NA;
                    const d2 x = bx[g & 1][j], t0 = b0[g & 1][j], t1 = b1[g & 1][j];
                    if constexpr (k < NA) { a0[q] = t0[0] * x[0]; a1[q] = t1[0] * x[0]; }
                    else { a0[q] = fma(t0[0], x[0], a0[q]); a1[q] = fma(t1[0], x[0], a1[q]); }
                    a0[q] = fma(t0[1], x[1], a0[q]); a1[q] = fma(t1[1], x[1], a1[q]);
                });
            };
#else
            constexpr int KQ = 2 * NTOP, NA = KQ < 7 ? KQ : 7, NG = (KQ + H - 1) / H;
            const d2 *xv = (const d2 *)(Xt + 4 * NTOP * (lane >> 4));
            double a[NA];
            d2 bx[2][H], b0[2][H];
            auto load = [&](auto gc) {
                constexpr int g = decltype(gc)::value, k0 = g * H, n = (k0 + H < KQ ? k0 + H : KQ) - k0;
                static_for<0, n>([&](auto jc) { constexpr int j = decltype(jc)::value, k = k0 + j; bx[g & 1][j] = xv[k]; b0[g & 1][j] = tm[k * 64]; });
            };
            auto compute = [&](auto gc) {
                constexpr int g = decltype(gc)::value, k0 = g * H, n = (k0 + H < KQ ? k0 + H : KQ) - k0;
                static_for<0, n>([&](auto jc) {
                    constexpr int j = decltype(jc)::value, k = k0 + j, q = k % NA;
                    const d2 x = bx[g & 1][j], t = b0[g & 1][j];
                    if constexpr (k < NA) a[q] = t[0] * x[0]; else a[q] = fma(t[0], x[0], a[q]);
                    a[q] = fma(t[1], x[1], a[q]);
                });
            };
#endif
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LATW_TOPV_PIPE) { load(std::integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0); }
            static_for<0, NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if constexpr (LATW_TOPV_PIPE) { if constexpr (g + 1 < NG) load(std::integral_constant<int, g + 1>{}); }
                else load(gc);
                __builtin_amdgcn_sched_barrier(0);
                compute(gc);
                __builtin_amdgcn_sched_barrier(0);
            });
#if LATW_TOP_VALU == 2
            double s0 = a0[0], s1 = a1[0];
            if constexpr (NA == 4) { s0 = (a0[0] + a0[1]) + (a0[2] + a0[3]); s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]); }
            else { static_for<1, NA>([&](auto jc) { constexpr int j = decltype(jc)::value; s0 += a0[j]; s1 += a1[j]; }); }
            // the eight column parts sit in eight neighbouring lanes: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror -- every lane of a
            // group ends with the same bits (each step adds the same two numbers on both sides)
            s0 += latw_dpp<0xB1>(s0); s1 += latw_dpp<0xB1>(s1);
            s0 += latw_dpp<0x4E>(s0); s1 += latw_dpp<0x4E>(s1);
            s0 += latw_dpp<0x141>(s0); s1 += latw_dpp<0x141>(s1);
            if ((lane & 7) == 0) *(d2 *)(Cc + latw_top_stage(r) * 16 + 2 * (lane >> 3)) = d2{s0, s1};
#else
            double s = a[0];
            if constexpr (NA == 7) s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + a[6]);
            else { static_for<1, NA>([&](auto jc) { s += a[decltype(jc)::value]; }); }
            s += lane_permute(s, 4 * (lane ^ 16));
            s += lane_permute(s, 4 * (lane ^ 32));
            if (lane < 16) Cc[latw_top_stage(r) * 16 + lane] = s;
#endif
        }
    });
}

// Tc <- K^-1 Tc.  All threads call; four barriers inside, none after the last phase (the caller's follows).
template <int N>
__device__ __forceinline__ void latw_solve(const d4 *fr, const double *TopL, const LatwVecs &v, double *Cc, int wv, int lane) {
    constexpr int NTOP = BcrFmt::top_count(N);
    static_assert(bcr_levels(N) >= 3 && NTOP >= 1, "two levels of reduction, then the dense top");
    if constexpr (!latw_fused(N)) {                       // (fused: done behind the right-hand side, the caller's barrier is this one)
        LATW_DISPATCH(wv, (latw_fwd<N, W, 0>(fr, v)))
        __syncthreads();
    }
    TICK(1)
    LATW_DISPATCH(wv, (latw_fwd<N, W, 1>(fr, v)))
    __syncthreads();
    TICK(2)
#if LATW_TOP_VALU
    LATW_DISPATCH(wv, (latw_top_valu<N, W>(TopL, TopL + LATW_TOP_LDS(N) - 32 * NTOP, Cc, lane)))
#else
    LATW_DISPATCH(wv, (latw_top<N, W>(TopL, v, lane)))
#endif
    __syncthreads();
    TICK(3)
    // level 1 back reads the top's solution from cb; on the way every top row's owner moves its stage into tb, where level 0 back (and the
    // owner passes) look for it -- nobody reads tb at a top stage during this phase
    if constexpr (latw_fused_back(N)) {
        LATW_DISPATCH(wv, (latw_bwd_own<N, W>(fr, v)))
        TICK(4)
    } else {
        LATW_DISPATCH(wv, (latw_bwd1<N, W>(fr, v)))
        __syncthreads();
        TICK(4)
        LATW_DISPATCH(wv, (latw_bwd<N, W, 0, false>(fr, v)))
    }
}

// ---- the round (as admm_lat, with ceil(NG / NWAVES) groups of four stages per wave) -----------------------------------------------------
enum { LATW_CONTINUE = 0, LATW_SOLVED = 1, LATW_GENERIC = 2 };
template <int NXT, int NUT, int NST>
__device__ int latw_check(double pv, double pv2, double ncq, double zA, double ysA, double omA, double zB, double ysB, double omB, double z0, double ys0, double om0,
                          double loA, double hiA, double loB, double hiB, double cc, int iter, int *frame_pin);      // (mpcqp_latw_check.h)

// FAST (iter0 >= 0): the round does not end with its iterations.  The termination test of OSQP (update_info + check_termination, mpcqp_phases.h:
// check_body) is evaluated right here in the OWNER layout -- A x and A'y are the two MFMA groups of the iteration applied to x and y, P x a 16-long
// dot product per lane against the weight matrices' LDS copy, the norms one block reduction (latw_check, a NON-INLINED function: compiled into this
// one, its temporaries and the 220 registers that live across it fought for the same file, and the spill reloads -- each a memory round trip --
// cost more than the generic check it replaced; as a call only the registers the callee really uses are saved around it) -- and
//   * converged: the solve is finished here (solution, iterate, mpcqp_info, statistics: what check_body's tail writes), return 1;
//   * not converged, and the cheap halves of both infeasibility certificates rule them out (|dy| or the support-function sum, |dx| or q'dx): the
//     next round starts at once, with the fragments and the owner registers where they are -- no write-back, no generic check, no reload;
//   * anything else (a rho estimate is due, the iteration limit, a certificate that needs its operator product, a non-finite residual, the first
//     launch of a two-launch solve): write-back and return 0 -- the generic check does exactly what it did before.
// Per round of one (12,4,30) instance this replaces write-back 1.5 k + generic check 15.2 k + owner registers 6.8 k + fragments 2.2 k cycles.
// The iteration count reached is left in Smem::iflag[5].
template <int NXT, int NUT, int NST>
__device__ __forceinline__ int admm_latw(const Lay &L, const HotPtrs &P, Smem &S, double *Xl, double *Zl, double *Yl, double alpha, int iters, int iter0) {
    constexpr int NB = 16, N = NST, NG = (N + 3) / 4, QN = (NG + NWAVES - 1) / NWAVES;
#ifdef MPCQP_RUN_TIMING
    const unsigned long long tf0_ = clock64();            // (development: the whole function on thread 0's clock, slot 11)
#endif
#ifndef LATW_FAST
#define LATW_FAST 1                // development: 0 = the in-function termination test compiled in but never taken; 2 = taken, but a converged solve is finished by the generic check
#endif
    const bool fast = LATW_FAST && iter0 >= 0 && L.hot_lds > L.hot_sz;      // (the check reads the weight matrices from their LDS copy)
    static_assert(NXT + NUT <= NB, "16 x 16 stages");
    const int nx = NXT ? NXT : L.nx, nu = NUT ? NUT : L.nu, NR = L.N;
    const int b = inst_of(P.perm), tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    gdouble *gx = (gdouble *)(P.x + (size_t)b * L.n), *gz = (gdouble *)(P.z + (size_t)b * L.m), *gy = (gdouble *)(P.y + (size_t)b * L.m);
    cgdouble *om = (cgdouble *)(P.omega + (size_t)b * L.m), *sv = (cgdouble *)(P.s + (size_t)b * L.n), *qv = (cgdouble *)S.Qv;
    gdouble *dxg = (gdouble *)(P.dx + (size_t)b * L.n), *dyg = (gdouble *)(P.dy + (size_t)b * L.m);
    const double cc = P.c[b], cinv = 1.0 / cc, beta = 1.0 - alpha;
    const double *hot = S.hot;
    constexpr int VS = LAT_VS(N);
    double *Tc = S.T + NB, *Cc = Tc + VS, *WA = Cc + VS, *WB = WA + VS;      // right-hand side / solution, c_e, W of the first / second row of a slot
    // the top inverse: behind the LDS copy of the iterate, on a 16-byte boundary
    double *TopL = Yl + L.m + ((smem_common_doubles(L) + L.n + 2 * L.m) & 1);      // (the dynamic LDS block itself starts on one)
    TICK_RESET
    TICK_START
#ifdef MPCQP_RUN_TIMING
    if (tid == 0) atomicAdd(&g_ticks[12], clock64() - tf0_);      // (development: what precedes the first tick)
#endif
    for (int i = tid; i < LAT_LDS_DOUBLES(N); i += NT) S.T[i] = 0.0;
    const double *Fb = P.F + (size_t)b * P.fsz;
    // The LDS-resident part of the factor -- the top inverse and the constant fragments G = [Ad Bd] (rows: dynamics rows, columns: (x, u)) and
    // G' -- outlives the round: nothing between two rounds touches that part of LDS except a refactorization, which says so (Smem::iflag[2],
    // cleared by factor_bcr and at kernel start).  98 KB per ROUND from L2 otherwise: 4 of a round's 7 microseconds of prologue.
    constexpr int NTOP = BcrFmt::top_count(N);
    if (S.iflag[2] == 0) {
        typedef __attribute__((address_space(1))) const d2 cgd2;
        cgd2 *Ft = (cgd2 *)(Fb + BcrFmt::top_off(N));
#if LATW_TOP_VALU
        cgd2 *Fv = (cgd2 *)(Fb + BcrFmt::topv_off(N));                      // (the copy factor_bcr left in this very order: 16 bytes per thread and trip, coalesced)
        {   // every trip's load requested before the first store: the stores go through a generic pointer, behind which the compiler keeps the next load --
            // thirteen dependent memory round trips (~ 12 us per kernel prologue, i.e. per queue item and per stepwise solve) instead of one
            constexpr int PAIRS = NTOP * NTOP * 128, TRIPS = (PAIRS + NT - 1) / NT;
            d2 buf[TRIPS];
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) { const int idx = tid + t * NT; buf[t] = Fv[idx < PAIRS ? idx : PAIRS - 1]; }
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) { const int idx = tid + t * NT; if (idx < PAIRS) *(d2 *)(TopL + 2 * idx) = buf[t]; }
        }
        (void)Ft;
#else
        for (int idx = tid; idx < NTOP * NTOP * 128; idx += NT) {           // (16 bytes per thread and trip: fragment element pairs (lane, j = 0,1 | 2,3))
            const int blk = idx >> 7, r = idx & 127, ln = r >> 1, hf = r & 1;
            *(d2 *)(TopL + ((blk * 2 + hf) * 64 + ln) * 2) = Ft[idx];
        }
#endif
        const double *Ad = hot + L.oAd, *Bd = hot + L.oBd;
        auto gent = [&](int r, int c) { return r < nx ? (c < nx ? Ad[r * nx + c] : (c < nx + nu ? Bd[r * nu + (c - nx)] : 0.0)) : 0.0; };
        if (wv == 0) {
            const d4 g = lat_make_frag(lane, gent);
            *(d2 *)(TopL + (((NTOP * NTOP) * 2 + 0) * 64 + lane) * 2) = d2{g[0], g[1]}; *(d2 *)(TopL + (((NTOP * NTOP) * 2 + 1) * 64 + lane) * 2) = d2{g[2], g[3]};
        } else if (wv == 1) {
            const d4 g = lat_make_frag(lane, [&](int r, int c) { return gent(c, r); });
            *(d2 *)(TopL + (((NTOP * NTOP + 1) * 2 + 0) * 64 + lane) * 2) = d2{g[0], g[1]}; *(d2 *)(TopL + (((NTOP * NTOP + 1) * 2 + 1) * 64 + lane) * 2) = d2{g[2], g[3]};
        }
        __syncthreads();
        if (tid == 0) S.iflag[2] = 1;
    }
    d4 fr[latw_max_slots(N)];
    const int lo16 = vec_lane_offset(lane);
    const int lI = lane >> 4, lB = (lane >> 2) & 3;
    const int o1 = 4 * ((lB + 1) & 3) + lI, o2 = 4 * ((lB + 2) & 3) + lI, o3 = 4 * ((lB + 3) & 3) + lI;
    const LatwVecs vec{Tc + lo16, Cc + lo16, Tc + o1, Tc + o2, Tc + o3, Cc + o1, Cc + o2, Cc + o3, TopL + LATW_TOP_LDS(N) - 32 * BcrFmt::top_count(N) + lo16, TopL + LATW_TOP_LDS(N) - 16 * BcrFmt::top_count(N) + lo16};
    // ---- owner map: lane (I, B, J), group g = wv + NWAVES q  ->  slot a = 4B + I of stage s = 4g + J
    const int a = 4 * ((lane >> 2) & 3) + (lane >> 4), J = lane & 3;
    const bool is_x = a < nx;
    const int jj = a - nx;
    const double cef = cc * hot[L.oeps];
    double pv[QN], pv2[QN], svp[QN], ncq[QN], sve[QN], kap[QN], okap[QN], te[QN], loA[QN], hiA[QN], loB[QN], hiB[QN];
    LatRow rA[QN], rB[QN], r0{0.0, 0.0, 0.0, 1.0};
    int sl[QN], pidx[QN], aidx[QN], bidx[QN];
    bool ok[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
        // (a group beyond the schedule's last owns nothing: it is parked on the zero slot behind the last stage)
        const int g = wv + NWAVES * q, s = g < NG ? 4 * g + J : 4 * NG;
        sl[q] = s * NB + a;
        ok[q] = is_x ? s < NR : (a < nx + nu && s < NR - 1);
        const int e = s * nx + a, cu = s * nu + jj;
        pidx[q] = is_x ? e : L.ou + cu; aidx[q] = is_x ? e : L.ri + cu; bidx[q] = is_x ? L.rs + e : L.rdu + nu + cu;
        te[q] = 0.0;
        // (every lane requests everything it may need, at clamped addresses, in one go -- the branches this replaces made a wave walk the state and
        //  the input lanes' loads one after the other, three memory round trips instead of one -- and keeps what applies to it)
        const bool live = ok[q], sx = live && is_x && L.soft;
        const int ip = live ? pidx[q] : 0, ia = live ? aidx[q] : 0, ib = live ? bidx[q] : 0, iq = live ? (is_x ? e : L.n_x + cu) : 0, ie = sx ? L.oe + e : 0;
        const double g_pv = gx[ip], g_sv = sv[ip], g_q = qv[iq], g_oA = om[ia], g_zA = gz[ia], g_yA = gy[ia], g_oB = om[ib], g_zB = gz[ib], g_yB = gy[ib];
        const double g_pe = gx[ie], g_se = sv[ie];
        const int jc = (jj >= 0 && jj < nu) ? jj : 0, ac = a < nx ? a : 0;
        const double b_x0 = S.x0s[ac], b_xlo = hot[L.oxmin + ac], b_xhi = hot[L.oxmax + ac];
        const double b_ulo = hot[L.oumin + jc], b_uhi = hot[L.oumax + jc], b_dlo = hot[L.oDumin + jc], b_dhi = hot[L.oDumax + jc];
        pv[q] = live ? g_pv : 0.0; svp[q] = live ? g_sv : 0.0; ncq[q] = live ? -cc * g_q : 0.0;
        rA[q] = LatRow{0.0, 0.0, 0.0, 1.0}; rB[q] = LatRow{0.0, 0.0, 0.0, 1.0};
        if (live) {
            rA[q].om = g_oA; rA[q].z = g_zA; rA[q].ys = cc * g_yA / g_oA; rA[q].w = g_oA * (g_zA - rA[q].ys);
            rB[q].om = g_oB; rB[q].z = g_zB; rB[q].ys = cc * g_yB / g_oB; rB[q].w = g_oB * (g_zB - rB[q].ys);
        }
        pv2[q] = sx ? g_pe : 0.0; sve[q] = sx ? g_se : 0.0;
        kap[q] = sx ? 1.0 / (cef + g_se + g_oB) : 0.0; okap[q] = sx ? g_oB * kap[q] : 0.0;
        loA[q] = live ? (is_x ? (s == 0 ? -b_x0 : 0.0) : b_ulo) : 0.0; hiA[q] = live ? (is_x ? (s == 0 ? -b_x0 : 0.0) : b_uhi) : 0.0;
        loB[q] = live ? (is_x ? b_xlo : b_dlo) : 0.0; hiB[q] = live ? (is_x ? b_xhi : b_dhi) : 0.0;
    }
    const bool u0v = wv == 0 && J == 0 && !is_x && a < nx + nu;      // the first-step rows u_0 - u_{-1} (mpc.py:574): stage 0's inputs
    {   // (requested by every lane, as above: no second round trip for wave 0 behind the branch)
        const int i0 = L.rdu + ((jj >= 0 && jj < nu) ? jj : 0);
        const double o0 = om[i0], z0 = gz[i0], y0 = gy[i0];
        if (u0v) { r0.om = o0; r0.z = z0; r0.ys = cc * y0 / o0; r0.w = o0 * (z0 - r0.ys); }
    }
    // neighbours in the flattened input sequence (mpc.py:570): slot a + 1 / a - 1, across the stage boundary at the ends
    const int onext = (jj + 1 < nu) ? 1 : NB - nu + 1, oprev = (jj > 0) ? -1 : -(NB - nu + 1);
    const double selx = is_x ? 1.0 : 0.0, sgnA = is_x ? -1.0 : 1.0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < QN; ++q) { WA[sl[q]] = rA[q].w; WB[sl[q]] = rB[q].w; }
    __syncthreads();
    TICK(8)
    int iter = iter0 >= 0 ? iter0 : 0, term = 0;
    for (;;) {
    // (the level fragments are (re)loaded at the start of EVERY round -- 2 k cycles from L2 -- so that they are dead during the termination test between two
    //  rounds: kept alive across it they leave the test no registers, and its spill reloads, each a memory round trip, cost several times the reload)
    TICK_START
    { const double *Fr = opaque_ptr(Fb); LATW_DISPATCH(wv, (latw_load<N, W>(Fr, fr))) }      // (opaque: a reload, not a loop-invariant the compiler may keep -- in scratch)
    TICK(7)
    for (int it = 1; it <= iters; ++it) {
        const bool keep_delta = it == iters;
        TICK_START
        // ---- right-hand side  s x - c q + A'W  with the slack eliminated; A'W = own rows' W + G' W_dyn of the next stage (MFMA, four stages a group)
        {
            double g[QN], h2[QN];
#pragma unroll
            for (int q = 0; q < QN; ++q) { g[q] = 0.0; h2[q] = 0.0; lat_mv(latw_top_frag(TopL, NTOP * NTOP + 1, lane), WA[sl[q] + NB], g[q], h2[q]); }
#pragma unroll
            for (int q = 0; q < QN; ++q) {
                const double wprev = WB[sl[q] + oprev];
                te[q] = fma(sve[q], pv2[q], rB[q].w) * kap[q];                     // (inputs and a hard state box: kap = 0, no slack)
                const double viaB = is_x ? fma(-rB[q].om, te[q], rB[q].w) : wprev - rB[q].w;
                double rhs = fma(svp[q], pv[q], ncq[q]) + (g[q] + h2[q]);
                rhs = fma(sgnA, rA[q].w, rhs) + viaB;
                if (q == 0) rhs += r0.w;
                Tc[sl[q]] = ok[q] ? rhs : 0.0;
            }
        }
        if constexpr (latw_fused(N)) { LATW_DISPATCH(wv, (latw_fwd0_own<N, W>(fr, vec))) }      // (level 0 forward of the wave's own stages: no barrier in between)
        __syncthreads();
        TICK(0)
        latw_solve<N>(fr, TopL, vec, Cc, wv, lane);
        __syncthreads();
        TICK(5)
        // ---- G v of the previous stage (MFMA), relaxation, projection, dual step of the owned rows
        {
            double g[QN], h2[QN];
#pragma unroll
            for (int q = 0; q < QN; ++q) { g[q] = 0.0; h2[q] = 0.0; lat_mv(latw_top_frag(TopL, NTOP * NTOP, lane), Tc[sl[q] - NB], g[q], h2[q]); }
#pragma unroll
            for (int q = 0; q < QN; ++q) {
                const double xt = Tc[sl[q]], un = Tc[sl[q] + onext];
                const double et = fma(-okap[q], xt, te[q]);
                const double vn = fma(alpha, xt, beta * pv[q]), en = fma(alpha, et, beta * pv2[q]);
                const double ztA = fma(selx, (g[q] + h2[q]) - xt - xt, xt);       // x: G v - xt;  u: ut
                const double ztB = is_x ? xt + et : un - xt;
                const double dA = lat_row_step(rA[q], ztA, loA[q], hiA[q], alpha, beta);
                const double dB = lat_row_step(rB[q], ztB, loB[q], hiB[q], alpha, beta);
                if (keep_delta && ok[q]) {
                    // (indices made opaque HERE: left to itself the compiler forms the five 64-bit addresses once, outside the iteration loop, keeps them
                    //  in scratch and reloads them in this branch -- ten registers' worth of spill traffic per round for five integer additions)
                    const int ip = opaque_lane(pidx[q]), ia = opaque_lane(aidx[q]), ib = opaque_lane(bidx[q]);
                    dxg[ip] = vn - pv[q];
                    if (is_x && L.soft) dxg[ip + L.oe] = en - pv2[q];
                    dyg[ia] = (rA[q].om * cinv) * dA; dyg[ib] = (rB[q].om * cinv) * dB;
                }
                pv[q] = vn; pv2[q] = en;
                if (q == 0 && u0v) { const double d0 = lat_row_step(r0, xt, S.du0[jj], S.du0[nu + jj], alpha, beta); if (keep_delta) dyg[opaque_lane(L.rdu + jj)] = (r0.om * cinv) * d0; }
                WA[sl[q]] = ok[q] ? rA[q].w : 0.0; WB[sl[q]] = ok[q] ? rB[q].w : 0.0;      // (a slot without a variable contributes nothing)
            }
        }
        __syncthreads();
        TICK(6)
    }
    iter += iters;
    if (!fast) break;
    TICK_START
    if constexpr (QN != 1) break;      // (several groups of stages per wave -- the 256-thread build: the generic check)
    else {
        // ---- the termination test, in a function of its own (latw_check below): the owned values travel BY VALUE, in registers
        const RunArgs &R = run_kargs().R;
        if (stop_mode(iter, R.max_iter, R.chk, R.rho_every, R.plain != 0) != COLD_CHECK) break;      // (rho estimate / iteration limit / plain iterations: the generic check)
        int verdict;
        { FramePin pin; verdict = latw_check<NXT, NUT, NST>(pv[0], pv2[0], ncq[0], rA[0].z, rA[0].ys, rA[0].om, rB[0].z, rB[0].ys, rB[0].om, r0.z, r0.ys, r0.om,
                                                               loA[0], hiA[0], loB[0], hiB[0], cc, iter, &pin.v); }
        verdict = __builtin_amdgcn_readfirstlane(verdict);
        if (verdict == LATW_SOLVED) { term = 1; break; }
        if (verdict != LATW_CONTINUE) break;
        iters = next_stop(iter, R.max_iter, R.chk, R.rho_every) - iter;
        // The next round starts from exactly the state a write-back and a reload would give it -- y leaves as ys (om / c) and comes back as c y / om:
        // the same two roundings here -- so that an instance's iterates do not depend on whether a round boundary was crossed in this function or through
        // the generic check (which launch structure a solve runs under -- device loop, one launch, two launches -- never changes a result).
        auto rt = [&](LatRow &r) { r.ys = cc * (r.ys * (r.om * cinv)) / r.om; r.w = r.om * (r.z - r.ys); };
        rt(rA[0]); rt(rB[0]);
        if (u0v) rt(r0);
        WA[sl[0]] = ok[0] ? rA[0].w : 0.0; WB[sl[0]] = ok[0] ? rB[0].w : 0.0;      // (a slot without a variable contributes nothing)
        __syncthreads();
    }
    TICK(9)
    }
    TICK_START
    if (!term) {
        // ---- end of the round: the iterate back to memory (global: next round / warm start; LDS copy: the residual evaluation)
        auto put_row = [&](const LatRow &r, int idx) { const double y = r.ys * (r.om * cinv); gz[idx] = r.z; gy[idx] = y; Zl[idx] = r.z; Yl[idx] = y; };
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            if (ok[q]) {
                gx[pidx[q]] = pv[q]; Xl[pidx[q]] = pv[q];
                if (is_x && L.soft) { gx[pidx[q] + L.oe] = pv2[q]; Xl[pidx[q] + L.oe] = pv2[q]; }
                put_row(rA[q], aidx[q]); put_row(rB[q], bidx[q]);
            }
        }
        if (u0v) put_row(r0, L.rdu + jj);
    }
    if (tid == 0) S.iflag[5] = iter;
    TICK(9)
    TICK_FLUSH
#ifdef MPCQP_RUN_TIMING
    if (tid == 0) atomicAdd(&g_ticks[11], clock64() - tf0_);
#endif
    return term;
}
