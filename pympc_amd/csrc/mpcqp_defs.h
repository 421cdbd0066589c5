// mpcqp_defs.h -- part of libmpcqp_hip: the constants both translation units share.
//   mpcqp.hip     host side, C ABI and every kernel with 256-thread workgroups (NT = 256, four waves: one per SIMD)
//   mpcqp_w8.hip  the same device headers compiled once more with NT = 512 (eight waves: two per SIMD) inside namespace w8, for the
//                 kernels of the latency backend at one instance per compute unit (MODE_BCRT, mpcqp_latw.h): a lone instance is bound by
//                 the stalls of its in-order waves -- dependent MFMAs (44 cycles), LDS round trips, double-precision vector
//                 instructions (32 cycles dependent) -- and a second wave on the same SIMD issues into them
#pragma once

#ifndef NT
#define NT 256                 // threads per workgroup (one workgroup = one MPC instance)
#endif
#define NWAVES (NT / 64)
#define QP_INFTY 1e30
#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4
#define RHO_MIN 1e-6
#define RHO_MAX 1e6
#define RHO_EQ_OVER_RHO_INEQ 1e3
#define RHO_TOL 1e-4

#ifdef MPCQP_LAT_ONLY
constexpr bool kLatOnly = true;    // this translation unit instantiates the latency kernels only: the other backends' code is parsed, never instantiated
#else
constexpr bool kLatOnly = false;
#endif
