// mpcqp_w8.hip -- second translation unit of libmpcqp_hip.so: the solve / closed-loop kernel of the latency backend with 512-THREAD workgroups
// (eight waves, two per SIMD) for handles with at most one instance per compute unit (mpcqp_create: Lay::nw = 8; mpcqp_latw.h says why).
//
// The device code of this library is written against the compile-time workgroup size NT (mpcqp_defs.h): here the same headers are compiled once
// more with NT = 512, inside namespace w8 so that nothing collides with the 256-thread instantiations of mpcqp.hip.  Only k_mpc_run of the
// cyclic-reduction modes with a dense top (MODE_BCRT + 11 / 21 / 31) is instantiated -- and, for ONE controller of the reference's cart pole on a
// long horizon (grouped stages, round staged in LDS), the sweeps kernel: half of its iteration is owner passes that eight waves run faster than
// four; their begin / check / factorization phases are the common ones at 512 threads.  Setup, export and the verification kernels of such a handle run from mpcqp.hip at 256 threads -- every phase sizes
// its loops by NT, and the data in memory does not know how many threads wrote it.
// Host side: one function, called by launch_run in mpcqp.hip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

#include "../../include/mpcqp.h"

#define NT 512
#define MPCQP_LAT_ONLY 1
#include "mpcqp_defs.h"

namespace w8 {
#include "mpcqp_layout.h"
#include "mpcqp_qp.h"
#include "mpcqp_factor.h"
#include "mpcqp_sweeps.h"
#include "mpcqp_group.h"
#include "mpcqp_bcr.h"
#include "mpcqp_wide.h"
#include "mpcqp_huge.h"
#include "mpcqp_dense.h"
#include "mpcqp_border.h"
#include "mpcqp_phases.h"
#include "mpcqp_run.h"
#include "mpcqp_tiny.h"
#include "mpcqp_lat.h"
#include "mpcqp_latw.h"
#include "mpcqp_kernels.h"
#include "mpcqp_latw_check.h"

template <int NXT, int NUT, int MODE, bool LOOP, bool LDSS = true>
static int launch(const RunKArgs &A, int grid, size_t smem, hipStream_t stream) {
    auto kernel = k_mpc_run<16, LDSS, NXT, NUT, MODE, LOOP>;
    if (smem > 48 * 1024 && hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return 1;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(NT), smem, stream, A);
    return 0;
}
template <int NXT, int NUT, int MODE, bool LDSS = true>
static int launch_lp(const RunKArgs &A, int loop, int grid, size_t smem, hipStream_t stream) {
    return loop ? launch<NXT, NUT, MODE, true, LDSS>(A, grid, smem, stream) : launch<NXT, NUT, MODE, false, LDSS>(A, grid, smem, stream);
}
}  // namespace w8

// sched: 11 / 21 / 31 = the cyclic-reduction schedule (spec12_4: the BASELINE shape with compile-time dimensions); 0 = ONE long-horizon controller
// on grouped stages with the round staged in LDS (the reference's cart pole -- its notebook and Kalman examples -- with compile-time dimensions,
// any other nx + nu <= 8 through the generic instantiation), spec12_4 then says "held input"
int mpcqp_w8_launch(const void *kargs, size_t kargs_bytes, int spec12_4, int sched, int loop, int grid, size_t smem, hipStream_t stream) {
    using namespace w8;
    if (kargs_bytes != sizeof(RunKArgs)) return 2;
    RunKArgs A; memcpy(&A, kargs, sizeof(A));
    if (sched == 0 && A.L.nx == 4 && A.L.nu == 1) return spec12_4 ? launch_lp<4, 1, MODE_BORDER, false>(A, loop, grid, smem, stream) : launch_lp<4, 1, MODE_CHAIN, false>(A, loop, grid, smem, stream);
    if (sched == 0) return spec12_4 ? launch_lp<0, 0, MODE_BORDER, false>(A, loop, grid, smem, stream) : launch_lp<0, 0, MODE_CHAIN, false>(A, loop, grid, smem, stream);
    if (sched == 31 && spec12_4) return launch_lp<12, 4, MODE_BCRT + 31>(A, loop, grid, smem, stream);
    if (sched == 31) return launch_lp<0, 0, MODE_BCRT + 31>(A, loop, grid, smem, stream);
    if (sched == 21) return launch_lp<0, 0, MODE_BCRT + 21>(A, loop, grid, smem, stream);
    if (sched == 11) return launch_lp<0, 0, MODE_BCRT + 11>(A, loop, grid, smem, stream);
    return 3;
}

#ifdef MPCQP_RUN_TIMING
// development build: this unit's phase clocks (mpcqp_get_stats prints them beside its own)
void mpcqp_w8_ticks(unsigned long long *out16) {
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(w8::g_ticks), 16 * sizeof(unsigned long long));
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(w8::g_ticks), z, sizeof(z));
}
#endif
