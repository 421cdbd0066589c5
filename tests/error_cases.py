"""Constructor inputs that pyMPC.mpc.MPCController rejects (mpc.py:82-223) or accepts in an unusual shape.
One table used by the golden generator (tests/golden/make_errors.py, which runs the REFERENCE class on every case and
records what happened) and by tests/test_ctor_errors.py (which demands the same from pympc_amd.MPCController)."""
import numpy as np


def base(nu=1):
    nx = 2
    Ad = np.array([[1.0, 0.2], [0.0, 0.97]])
    Bd = np.array([[0.0], [0.1]]) if nu == 1 else np.array([[0.0, 0.1], [0.1, -0.05]])
    return dict(Ad=Ad, Bd=Bd, Np=5, x0=np.array([0.1, 0.2]), xref=np.array([1.0, 0.0]), uref=np.zeros(nu), uminus1=np.zeros(nu),
                Qx=np.diag([0.5, 0.1]), QxN=np.diag([0.5, 0.1]), Qu=2.0 * np.eye(nu), QDu=10.0 * np.eye(nu),
                xmin=-np.ones(nx), xmax=np.ones(nx), umin=-np.ones(nu), umax=np.ones(nu), Dumin=-0.2 * np.ones(nu), Dumax=0.2 * np.ones(nu))


def _set(**changes):
    def make(nu=1):
        kw = base(nu)
        for k, v in changes.items():
            if v is _DROP:
                kw.pop(k)
            else:
                kw[k] = v
        return kw
    return make


_DROP = object()

CASES = {
    'Ad_not_square': _set(Ad=np.ones((2, 3))),
    'Ad_1d': _set(Ad=np.ones(4)),
    'Bd_wrong_rows': _set(Bd=np.ones((3, 1))),
    'Bd_1d': _set(Bd=np.ones(2)),
    'Np_1': _set(Np=1),
    'Nc_gt_Np': _set(Nc=6),
    'Nc_eq_Np': _set(Nc=5),
    'x0_wrong_size': _set(x0=np.ones(3)),
    'x0_column': _set(x0=np.ones((2, 1))),
    'x0_row': _set(x0=np.ones((1, 2))),
    'xref_wrong_size': _set(xref=np.ones(3)),
    'xref_2d_wrong_cols': _set(xref=np.ones((6, 3))),
    'xref_2d_too_few_rows': _set(xref=np.ones((4, 2))),
    'xref_2d_Np_rows': _set(xref=np.ones((5, 2))),
    'xref_2d_Np1_rows': _set(xref=np.ones((6, 2))),
    'xref_row': _set(xref=np.ones((1, 2))),
    'uref_wrong_size': _set(uref=np.ones(2)),
    'uminus1_wrong_size': _set(uminus1=np.ones(2)),
    'uminus1_default_aliases_uref': _set(uminus1=_DROP),
    'Qx_wrong_shape': _set(Qx=np.eye(3)),
    'Qx_1d': _set(Qx=np.ones(2)),
    'QxN_wrong_rows': _set(QxN=np.eye(3)),
    'QxN_wrong_cols_only': _set(QxN=np.ones((2, 3))),
    'QxN_without_Qx': _set(Qx=_DROP),
    'Qu_wrong_shape': _set(Qu=np.eye(2)),
    'QDu_wrong_shape': _set(QDu=np.eye(2)),
    'xmin_wrong_size': _set(xmin=-np.ones(3)),
    'xmax_wrong_size': _set(xmax=np.ones(3)),
    'xmax_column': _set(xmax=np.ones((2, 1))),
    'umin_wrong_size': _set(umin=-np.ones(2)),
    'umax_wrong_size': _set(umax=np.ones(2)),
    'Dumin_wrong_size': _set(Dumin=-np.ones(2)),
    'Dumax_wrong_size': _set(Dumax=np.ones(2)),
    'all_defaults': lambda nu=1: dict(Ad=base()['Ad'], Bd=base()['Bd']),
    'scipy_sparse_matrices': lambda nu=1: dict(base(), Ad=__import__('scipy.sparse', fromlist=['x']).csc_matrix(base()['Ad']),
                                               Qx=__import__('scipy.sparse', fromlist=['x']).diags([0.5, 0.1])),
}

# attributes whose stored shape is compared when the reference accepts the input
SHAPED = ('x0', 'xref', 'uref', 'uminus1', 'Qx', 'QxN', 'Qu', 'QDu', 'xmin', 'xmax', 'umin', 'umax', 'Dumin', 'Dumax')


def outcome(Ctrl, make):
    """What constructing ``Ctrl(**make())`` does: ('error', exception type name, message) or ('ok', {attr: shape}, Nc,
    whether uminus1 aliases uref)."""
    kw = make()
    try:
        K = Ctrl(**kw)
    except Exception as e:            # noqa: BLE001 -- the exception type is part of what is compared
        return ['error', type(e).__name__, str(e)]
    return ['ok', {a: list(np.shape(getattr(K, a))) for a in SHAPED}, int(K.Nc), bool(K.uminus1 is K.uref)]
