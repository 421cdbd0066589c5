// mpcqp_topv.h -- part of libmpcqp_hip (included by mpcqp_bcr.h): the order the dense top of the cyclic reduction is stored in for the round's vector-ALU mat-vec,
// as plain host / device functions (tests/test_topv_order.py compiles them with g++ and checks that the two maps are inverse to each other).
#pragma once
#ifndef __host__
#define __host__
#define __device__
#endif
// The dense top of the round on the VECTOR ALU (mpcqp_latw.h says why and what was measured): 0 = on the matrix cores (fragments), 1 = lane (i = lane & 15,
// p = lane >> 4) of the wave that owns block row r holds row 16 r + i times columns [4 nt p, 4 nt (p + 1)), 2 = lane (q = lane >> 3, p = lane & 7) rows 16 r + 2 q,
// 16 r + 2 q + 1 times columns [2 nt p, 2 nt (p + 1)).  Order of the inverse in 16-byte pairs: pair ((r * 2 nt + k) * 64 + lane), k < 2 nt, holds
//   mode 1: T[16 r + (lane & 15)][4 nt (lane >> 4) + 2 k .. + 1]          mode 2: T[16 r + 2 (lane >> 3) + k / nt][2 nt (lane & 7) + 2 (k % nt) .. + 1]
// -- a wave's read k is 64 consecutive pairs (conflict-free).  bcr_topv_rc: (row, first column) of a pair.
#ifndef LATW_TOP_VALU
#define LATW_TOP_VALU 2
#endif
__host__ __device__ inline void bcr_topv_rc(int nt, int pair, int &row, int &col) {
    const int ln = pair & 63, rk = pair >> 6, r = rk / (2 * nt), k = rk - r * 2 * nt;
#if LATW_TOP_VALU == 2
    const int rr = k / nt, kc = k - rr * nt;
    row = 16 * r + 2 * (ln >> 3) + rr; col = 2 * nt * (ln & 7) + 2 * kc;
#else
    row = 16 * r + (ln & 15); col = 4 * nt * (ln >> 4) + 2 * k;
#endif
}

// ... and the position (in doubles) of entry (row, col) in that order
__host__ __device__ inline long long bcr_topv_pos(int nt, int row, int col) {
    const int r = row >> 4, rl = row & 15;
#if LATW_TOP_VALU == 2
    const int ln = (rl >> 1) * 8 + col / (2 * nt), k = (rl & 1) * nt + (col % (2 * nt)) / 2;
#else
    const int ln = rl + 16 * (col / (4 * nt)), k = (col % (4 * nt)) / 2;
#endif
    return 2LL * ((long long)(r * 2 * nt + k) * 64 + ln) + (col & 1);
}

