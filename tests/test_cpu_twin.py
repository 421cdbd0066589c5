"""The product's Python host layer -- MPCController (the drop-in class), DeviceProblem (the osqp-shaped seam), BatchMPCController
(stepwise and the closed loop behind mpcqp_mpc_loop), the CSC seam, mpcqp_step_host -- driven through the SAME C ABI
(include/mpcqp.h) answered on the CPU by oracle/libmpcqp_cpu.so (the oracle behind a C restatement of the reference's QP
builder; test infrastructure).  Here, without a GPU, this exercises every ctypes call, array conversion, lazy update and
status path the GPU run takes, against the reference-made goldens; on the GPU the same Python runs against libmpcqp_hip.so
(tests/test_closed_loop_golden.py, tests/test_gpu_seam.py, tests/test_gpu_loop_parity.py)."""
import os
import subprocess
import warnings

import numpy as np
import pytest

from util import golden_names, load_golden, golden_kwargs, golden_csc, update_steps, apply_attrs, traj_names, load_traj, cart_pole_plant
from test_closed_loop_golden import _closed_loop, _no_slack_controller, _no_slack_loop

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOFT = [n for n in golden_names() if not n.endswith('_hard')]


@pytest.fixture
def twin():
    """pympc_amd bound to the CPU twin for one test (the product loader itself reads no switch: the test swaps the path)."""
    from pympc_amd import _lib
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle'), 'libmpcqp_cpu.so'])
    old = (_lib.LIB_PATH, _lib._lib)
    _lib.LIB_PATH, _lib._lib = os.path.join(ROOT, 'oracle', 'libmpcqp_cpu.so'), None
    try:
        yield _lib.load()
    finally:
        _lib.LIB_PATH, _lib._lib = old


def test_twin_exports_the_whole_abi(twin):
    from pympc_amd import _lib
    assert all(hasattr(twin, s) for s in _lib.SYMBOLS)


@pytest.mark.parametrize('name', traj_names())
def test_drop_in_class_through_the_abi_reproduces_reference_closed_loop(twin, name):
    from pympc_amd import MPCController, fixtures
    g = load_traj(name)
    kw = dict(fixtures.NAMED[str(g['fixture'])]())
    kw.update(eps_abs=float(g['eps']), eps_rel=float(g['eps']))
    K = MPCController(**kw)
    K.solver_settings = dict(max_iter=400000)
    Ad, Bd = kw['Ad'], kw['Bd']
    plant = cart_pole_plant if name == 'cart_pole' else (lambda x, u: Ad @ x + Bd @ u)
    _closed_loop(K, kw, g, plant, feed_golden_states=False, tol=1e-7)
    assert type(K.prob).__name__ == 'DeviceProblem'


def test_no_slack_shim_through_the_abi(twin):
    g = load_traj('no_slack_point_mass')
    _no_slack_loop(_no_slack_controller(g), g, 1e-7)


@pytest.mark.parametrize('name', golden_names())
def test_twin_builds_the_reference_qp(twin, name):
    """mpcqp_setup + mpcqp_export_qp of the twin = the C restatement of mpc.py:456-608 against the reference-built matrices."""
    import scipy.sparse as sp
    g = load_golden(name)
    K = apply_attrs(__import__('pympc_amd').MPCController(**golden_kwargs(g)), golden_kwargs(g))
    K.setup(solve=False)
    P, q, A, l, u = K.prob.batch_problem.export_qp()
    U = sp.triu(golden_csc(g, 'P')).toarray()
    assert np.array_equal(P[0], U + np.triu(U, 1).T)
    assert np.array_equal(A[0], golden_csc(g, 'A').toarray())
    assert np.allclose(q[0], g['q'], rtol=2e-15, atol=1e-300)
    assert np.array_equal(l[0], np.clip(g['l'], -1e30, 1e30)) and np.array_equal(u[0], np.clip(g['u'], -1e30, 1e30))
    for st in update_steps(g):
        K.update(st['x'], u=st['u'], xref=st['xref'], solve=False)
        _, q, _, l, u = K.prob.batch_problem.export_qp()
        assert np.allclose(q[0], st['q'], rtol=2e-15, atol=1e-300)
        assert np.array_equal(l[0], np.clip(st['l'], -1e30, 1e30)) and np.array_equal(u[0], np.clip(st['u_bound'], -1e30, 1e30))
        if 'upd0_output_u' in g.files and np.array_equal(st['q'], g['upd0_q']):
            K.uminus1_rh = np.array(g['upd0_output_u'])


@pytest.mark.parametrize('name', SOFT[:6])
def test_matrix_seam_through_the_abi(twin, name):
    """DeviceProblem.setup(P, q, A, l, u) / update(q=, l=, u=) / solve(): mpcqp_create_csc, mpcqp_setup_csc, mpcqp_update_vectors."""
    from pympc_amd.solver import DeviceProblem
    from oracle.osqp_oracle import OSQP
    g = load_golden(name)
    P, A = golden_csc(g, 'P'), golden_csc(g, 'A')
    # (the twin takes any QP: it has no controller dimensions to report -- DeviceProblem asks for them, so bind the raw calls)
    from pympc_amd.solver import BatchProblem
    bp = BatchProblem.__new__(BatchProblem)
    import ctypes as C
    from pympc_amd import _lib
    from pympc_amd.solver import make_settings, _ptr
    bp._L, bp._h, bp.batch = twin, C.c_void_p(), 1
    bp.settings = make_settings(eps_abs=1e-9, eps_rel=1e-9, max_iter=200000)
    Pc, Ac = P.tocsc(), A.tocsc(); Pc.sort_indices(); Ac.sort_indices()
    arrs = [np.ascontiguousarray(Pc.indptr, dtype=np.int64), np.ascontiguousarray(Pc.indices, dtype=np.int32),
            np.ascontiguousarray(Ac.indptr, dtype=np.int64), np.ascontiguousarray(Ac.indices, dtype=np.int32)]
    _lib.check(twin.mpcqp_create_csc(C.byref(bp._h), 0, 1, Pc.shape[0], Ac.shape[0], *[_ptr(a) for a in arrs], 0, 0, C.byref(bp.settings)), 'mpcqp_create_csc')
    bp._csc = (Pc, Ac); bp.n, bp.m = Pc.shape[0], Ac.shape[0]; bp.nx = bp.nu = 1
    bp.setup_csc(Pc.data[None], Ac.data[None], np.asarray(g['q'])[None], np.asarray(g['l'])[None], np.asarray(g['u'])[None])
    O = OSQP(); O.setup(P, g['q'], A, g['l'], g['u'], eps_abs=1e-9, eps_rel=1e-9, max_iter=200000)
    bp.solve_async(); x, y, info = bp.solution(); ro = O.solve()
    assert info[0].status == 1 and ro.info.status == 'solved'
    assert np.abs(x[0] - ro.x).max() <= 1e-6 * max(1.0, np.abs(ro.x).max())
    for st in update_steps(g):
        bp.update_vectors(np.asarray(st['q'])[None], np.clip(st['l'], -1e30, 1e30)[None], np.clip(st['u_bound'], -1e30, 1e30)[None])
        O.update(q=st['q'], l=st['l'], u=st['u_bound'])
        bp.solve_async(); x, y, info = bp.solution(); ro = O.solve()
        assert np.abs(x[0] - ro.x).max() <= 1e-6 * max(1.0, np.abs(ro.x).max())
    bp.close()


def test_batch_controller_stepwise_equals_closed_loop_call(twin):
    """BatchMPCController.step() per step against .run() (mpcqp_mpc_loop): same inputs, states, statuses."""
    from pympc_amd import BatchMPCController, fixtures
    B = 3
    kws = [fixtures.random_lti(20 + i, nx=5, nu=3, Np=8) for i in range(B)]
    st = lambda k: np.stack([np.asarray(kw[k], dtype=float) for kw in kws])
    mk = lambda: BatchMPCController(st('Ad'), st('Bd'), Np=8, x0=st('x0'), Qx=st('Qx'), QxN=st('QxN'), Qu=st('Qu'), QDu=st('QDu'), xmin=st('xmin'), xmax=st('xmax'),
                                    umin=st('umin'), umax=st('umax'), Dumin=st('Dumin'), Dumax=st('Dumax'), eps_feas=1e6, eps_abs=1e-8, eps_rel=1e-8, max_iter=100000)
    rng = np.random.default_rng(0)
    w = 0.01 * rng.standard_normal((6, B, 5))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Ka = mk(); Ka.setup(); tr = Ka.run(6, w=w)
        Kb = mk(); Kb.setup()
        x = st('x0')
        for k in range(6):
            u = Kb.output()
            assert np.abs(u - tr['u'][k]).max() <= 1e-9 * max(1.0, np.abs(tr['u']).max())
            assert np.abs(x - tr['x'][k]).max() <= 1e-9
            x = np.einsum('bij,bj->bi', st('Ad'), x) + np.einsum('bij,bj->bi', st('Bd'), u) + w[k]
            Kb.update(x)
    assert (tr['status'] == 1).all()


def test_unconstrained_gains_host_logic_through_the_twin(twin):
    """pympc_amd/unconstrained.py (unit-vector batch, mpcqp_eq_solve sweeps until the gains stop moving, slicing) through the ABI, the
    twin's sweeps being the oracle's ADMM iterations on the same equality-constrained problems: against the condensed closed form."""
    from pympc_amd import MPCController, fixtures
    from closed_form import unconstrained_mpc
    kw = fixtures.point_mass()
    K = MPCController(**kw)
    G = K.unconstrained_gains()
    nx, nu, Np = 2, 1, kw['Np']
    for which, key, size in (('x0', 'K_x0', nx), ('xref', 'K_xref', nx), ('uref', 'K_uref', nu), ('uminus1', 'K_um1', nu)):
        for j in range(size):
            args = dict(x0=np.zeros(nx), xref=np.zeros(nx), uref=np.zeros(nu), uminus1=np.zeros(nu))
            args[which] = np.eye(size)[j]
            u_seq, _ = unconstrained_mpc(kw['Ad'], kw['Bd'], Np, Qx=kw['Qx'], QxN=kw.get('QxN'), Qu=kw['Qu'], QDu=kw['QDu'], **args)
            assert np.abs(G[key][:, j] - u_seq.ravel()).max() <= 1e-8 * max(1.0, np.abs(u_seq).max()), (key, j)


def test_batch_host_layer_one_model_many_states_through_the_abi(twin):
    """BatchMPCController.share_factor() (mpcqp_share_factor) through the ABI: the twin keeps a factor per instance and says so (0 share), and the
    one-model-many-states call order -- setup with ONE common state, share, scatter the states -- gives each state the controller's own answer."""
    from pympc_amd import BatchMPCController, MPCController, fixtures
    kw = fixtures.random_lti(2, nx=4, nu=2, Np=6)
    B = 5
    st = lambda a: np.broadcast_to(np.asarray(a, dtype=float), (B,) + np.shape(a))
    rng = np.random.default_rng(0)
    X, Um1 = rng.standard_normal((B, 4)), 0.1 * rng.standard_normal((B, 2))
    K = BatchMPCController(st(kw['Ad']), st(kw['Bd']), Np=kw['Np'], x0=st(kw['x0']), uminus1=st(kw['uminus1']), xref=st(kw['xref']), uref=st(kw['uref']),
                           Qx=st(kw['Qx']), QxN=st(kw['QxN']), Qu=st(kw['Qu']), QDu=st(kw['QDu']), xmin=st(kw['xmin']), xmax=st(kw['xmax']),
                           umin=st(kw['umin']), umax=st(kw['umax']), Dumin=st(kw['Dumin']), Dumax=st(kw['Dumax']), eps_abs=1e-9, eps_rel=1e-9)
    K.solver_settings = dict(max_iter=100000)
    K.setup()
    assert K.share_factor() == 0
    K.update(X, Um1)
    U = K.output()
    for i in range(B):
        K1 = MPCController(**dict(kw, x0=X[i], uminus1=Um1[i], eps_abs=1e-9, eps_rel=1e-9)); K1.solver_settings = dict(max_iter=100000); K1.setup()
        assert np.abs(U[i] - K1.output()).max() <= 1e-6
