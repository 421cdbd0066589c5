"""mpcqp_create_csc / mpcqp_setup_csc -- the solver seam with the caller's matrices behind the C ABI (pyMPC/mpc.py:266).
CPU part: the pattern analysis runs before any GPU is touched, so a matrix pair that is not pyMPC's must come back as
MPCQP_ERR_UNSUPPORTED (-4) and pyMPC's own patterns must get as far as the device (-3 = no GPU here).  GPU part: the
reference-built matrices of every golden fixture go through the C path (values recovered and rebuilt IN the library) and the
solve must match the oracle's; tampered values must be refused."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from util import golden_names, load_golden, golden_csc, golden_kwargs

ALL = golden_names()          # (SOFT_ON = True and the *_hard fixtures of SOFT_ON = False: the pattern tells them apart)


def _create(P, A, nx=0, nu=0):
    from pympc_amd import _lib
    L = _lib.load()
    Pc, Ac = sp.csc_matrix(P), sp.csc_matrix(A)
    Pc.sort_indices(); Ac.sort_indices()
    h = C.c_void_p()
    arrs = [np.ascontiguousarray(Pc.indptr, dtype=np.int64), np.ascontiguousarray(Pc.indices, dtype=np.int32),
            np.ascontiguousarray(Ac.indptr, dtype=np.int64), np.ascontiguousarray(Ac.indices, dtype=np.int32)]
    rc = L.mpcqp_create_csc(C.byref(h), 0, 1, Pc.shape[0], Ac.shape[0], *[a.ctypes.data_as(C.c_void_p) for a in arrs], nx, nu, None)
    msg = L.mpcqp_last_error().decode()
    if rc == 0:
        L.mpcqp_destroy(h)
    return rc, msg


@pytest.mark.parametrize('name', ALL)
def test_pattern_of_every_reference_qp_is_accepted(name):
    g = load_golden(name)
    rc, msg = _create(golden_csc(g, 'P'), golden_csc(g, 'A'))
    assert rc in (0, -3), (rc, msg)            # accepted: created (GPU box) or stopped at 'no HIP device' (here)


def test_foreign_patterns_are_refused_before_any_gpu_work():
    rng = np.random.default_rng(3)
    A = sp.random(40, 30, density=0.2, random_state=1, format='csc'); P = sp.eye(30, format='csc')
    rc, msg = _create(P, A)
    assert rc == -4 and 'not an MPC QP' in msg
    g = load_golden('point_mass')
    A = golden_csc(g, 'A').tolil(); A = sp.vstack([A, sp.csr_matrix(np.ones((1, A.shape[1])))]).tocsc()      # one more row
    rc, msg = _create(golden_csc(g, 'P'), A)
    assert rc == -4
    assert rng is not None


@pytest.mark.gpu
@pytest.mark.parametrize('name', ALL)
def test_reference_matrices_through_the_c_seam(name):
    """DeviceProblem.setup(P, q, A, l, u) = mpcqp_create_csc + mpcqp_setup_csc: same solve as the oracle on the same matrices."""
    from pympc_amd.solver import DeviceProblem
    from oracle.osqp_oracle import OSQP
    g = load_golden(name)
    P, A = golden_csc(g, 'P'), golden_csc(g, 'A')
    kw = golden_kwargs(g)
    D = DeviceProblem(); D.setup(P, g['q'], A, g['l'], g['u'], eps_abs=1e-9, eps_rel=1e-9, max_iter=200000)
    bp = D.batch_problem
    assert (bp.nx, bp.Np) == (np.asarray(kw['Ad']).shape[0], kw['Np'])
    r = D.solve()
    O = OSQP(); O.setup(P, g['q'], A, g['l'], g['u'], eps_abs=1e-9, eps_rel=1e-9, max_iter=200000)
    ro = O.solve()
    assert r.info.status == ro.info.status == 'solved'
    assert np.abs(r.x - ro.x).max() <= 1e-6 * max(1.0, np.abs(ro.x).max())


@pytest.mark.gpu
def test_tampered_values_are_refused_by_the_library():
    from pympc_amd.solver import DeviceProblem
    from pympc_amd.qp_recover import NotAnMPCQP
    g = load_golden('cart_pole')
    P, A = golden_csc(g, 'P').tocsc(), golden_csc(g, 'A').tocsc()
    A2 = A.copy(); A2.data[len(A2.data) // 2] *= 1.0000001                     # a stage whose Ad differs from the others
    with pytest.raises(NotAnMPCQP):
        DeviceProblem().setup(P, g['q'], A2, g['l'], g['u'])
    l2 = np.array(g['l']); l2[-1] += 1.0                                       # a Delta-u bound that is not stage-periodic
    with pytest.raises(NotAnMPCQP):
        DeviceProblem().setup(P, g['q'], A, l2, g['u'])
    q2 = np.array(g['q']); q2[-1] = 1.0                                        # a cost on a slack variable
    with pytest.raises(NotAnMPCQP):
        DeviceProblem().setup(P, q2, A, g['l'], g['u'])


@pytest.mark.gpu
@pytest.mark.parametrize('Np,Nc,Qu,QDu,nu', [(150, 75, 0.1, 0.3, 1), (20, 5, 3.3, 0.7, 1), (12, 3, 0.1, 0.3, 2)])
def test_held_input_with_inexact_weights_through_the_c_seam(Np, Nc, Qu, QDu, nu):
    """ADVICE r3 (mpcqp_csc.h): Nc < Np with input weights that are not exactly representable -- the library's rebuild-and-compare
    accepts the genuine pyMPC matrices (the last input block carries Np - Nc + 1 times the rounding error of Qu) and solves them."""
    from pympc_amd.solver import DeviceProblem
    from oracle.osqp_oracle import OSQP
    from test_qp_recover import _held_input_qp
    K = _held_input_qp(Np, Nc, Qu, QDu, nu)
    D = DeviceProblem(); D.setup(K.P, K.q, K.A, K.l, K.u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000)
    r = D.solve()
    O = OSQP(); O.setup(K.P, K.q, K.A, K.l, K.u, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000)
    ro = O.solve()
    assert r.info.status == ro.info.status == 'solved'
    assert np.abs(r.x - ro.x).max() <= 1e-6 * max(1.0, np.abs(ro.x).max())


def test_bad_column_pointers_are_an_argument_error():
    """non-monotone or out-of-range column pointers must come back as MPCQP_ERR_ARG, not as an out-of-bounds read."""
    from pympc_amd import _lib
    L = _lib.load()
    g = load_golden('point_mass')
    Pc, Ac = sp.csc_matrix(golden_csc(g, 'P')), sp.csc_matrix(golden_csc(g, 'A'))
    Pc.sort_indices(); Ac.sort_indices()
    for which in ('P', 'A'):
        pp, ap = np.array(Pc.indptr, dtype=np.int64), np.array(Ac.indptr, dtype=np.int64)
        bad = pp if which == 'P' else ap
        bad[3] = bad[-1] + 5                           # an interior pointer beyond colptr[n]
        arrs = [pp, np.ascontiguousarray(Pc.indices, dtype=np.int32), ap, np.ascontiguousarray(Ac.indices, dtype=np.int32)]
        h = C.c_void_p()
        rc = L.mpcqp_create_csc(C.byref(h), 0, 1, Pc.shape[0], Ac.shape[0], *[a.ctypes.data_as(C.c_void_p) for a in arrs], 0, 0, None)
        assert rc == -1, (which, rc, L.mpcqp_last_error().decode())
