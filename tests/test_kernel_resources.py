"""The design counts on resident workgroups per compute unit: four for 16 x 16 stages (128 VGPRs: the bandwidth kernel of the
headline batch), two for 32 x 32 stages, one for the latency kernels (register-resident factor).  A shared non-inlined phase
once inherited the one-per-CU register budget and silently halved the headline throughput -- so the compiler's own
resource remarks of the build (pympc_amd/libmpcqp_hip.kernel_resources.txt, written by csrc/build.sh) are checked."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, 'pympc_amd', 'libmpcqp_hip.kernel_resources.txt')


def _kernels():
    if not os.path.exists(PATH):
        pytest.skip('no kernel_resources.txt next to the library (built without csrc/build.sh)')
    out, cur = {}, None
    for line in open(PATH):
        m = re.match(r'\s*Function Name: (\S+)', line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.match(r'\s*([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)', line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return out


def _run_kernels(ks):
    """k_mpc_run<NB, LDSSTATE, NXT, NUT, MODE, LOOP> instantiations by template arguments."""
    res = {}
    for name, v in ks.items():
        m = re.match(r'_Z9k_mpc_runILi(\d+)ELb([01])ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])EE', name)
        if m:
            res[tuple(int(g) for g in m.groups())] = v
    return res


def test_occupancy_of_the_solve_kernels():
    rk = _run_kernels(_kernels())
    assert rk, 'no k_mpc_run instantiation in the resource remarks'
    for (nb, lds, nxt, nut, mode, loop), v in rk.items():
        vg, occ = int(v['VGPRs']), int(v['Occupancy'])
        if mode in (0, 1) and nb == 16:
            assert vg <= 128 and occ >= 4, ('16 x 16 sweeps must keep four workgroups per CU', nb, mode, loop, v)
        elif mode in (0, 1) and nb == 32:
            assert vg + int(v['AGPRs']) <= 256 and occ >= 2, ('32 x 32 sweeps must keep two workgroups per CU', nb, mode, loop, v)
        else:
            assert occ >= 1
    assert (16, 1, 12, 4, 0, 1) in rk and (32, 0, 20, 8, 0, 1) in rk          # the BASELINE specialisations exist


def test_the_512_thread_latency_kernels_fit_two_waves_per_simd():
    """mpcqp_w8.hip: eight waves per workgroup share a compute unit's four SIMDs two by two, so a wave has 256 registers -- the kernels must
    be allocated for exactly that (occupancy 2), and the scratch the whole call tree asks for stays small (the hot loop itself is
    spill-free: the factorization phase is what uses it)."""
    ks = _kernels()
    w8 = {n: v for n, v in ks.items() if n.startswith('_ZN2w89k_mpc_runILi16E')}
    # cyclic reduction: (12,4) and generic at 31 stages, generic at 21 and 11; grouped stages (the reference cart pole and generic), with and without a held
    # input; solve and closed loop of each
    assert len(w8) == 16, sorted(w8)
    for n, v in w8.items():
        assert int(v['Occupancy']) == 2 and int(v['VGPRs']) + int(v['AGPRs']) <= 256, (n, v)
        assert int(v['ScratchSize']) <= 512, (n, v)


def test_scratch_of_the_hot_kernels_stays_where_it_was():
    """Correctness and speed of these kernels lean on compiler behaviour that is not specified (phases without callee-saved registers through
    a pinned caller-frame argument, every device function force-inlined): a toolchain that changes it shows up here first -- as scratch.  Values
    seen with ROCm 7.2 (bytes per lane over the kernel's call tree; the ADMM phases themselves use 6-10 scratch instructions, SGPR spill lanes):
    628 / 500 for the four-per-CU kernels (their factorization phase), 68 for 32 x 32 stages, 244 / 116 for the 512-thread latency kernels,
    68 for the 512-thread cart-pole kernels."""
    ks = _kernels()
    rk = _run_kernels(ks)
    # (round 6: the kernel's own frame grew by ~ 60 bytes -- the persistent loop around the instance -- 780 / 668 for the four-per-CU kernels, 316 / 204 for the
    #  512-thread latency kernels, whose ADMM phase itself went from 148 to 80 bytes)
    lim = {(16, 1, 12, 4, 0, 1): 832, (16, 1, 12, 4, 0, 0): 704, (32, 0, 20, 8, 0, 1): 192, (32, 0, 20, 8, 0, 0): 192}
    for key, bound in lim.items():
        assert int(rk[key]['ScratchSize']) <= bound, (key, rk[key])
    for n, v in ks.items():
        if n.startswith('_ZN2w89k_mpc_runILi16ELb1ELi12ELi4ELi231E'):
            assert int(v['ScratchSize']) <= 320, (n, v)
        if n.startswith('_ZN2w89k_mpc_runILi16ELb0ELi4ELi1E'):
            assert int(v['ScratchSize']) <= 128, (n, v)
