"""Drop-in replacement for ``pyMPC.mpc.MPCController`` (reference: pyMPC/mpc.py:27-616).

Same constructor signature, defaults, error messages, public attributes and
``setup / update / solve / output / __controller_function__`` semantics.  The difference is
what sits behind ``self.prob``: instead of ``osqp.OSQP()`` (mpc.py:241) it is a
``pympc_amd.solver.DeviceProblem`` -- a ctypes handle onto ``libmpcqp_hip.so``, the
hand-written HIP implementation of the QP build and of the OSQP-style ADMM loop for gfx950.

The hidden switch ``SOFT_ON = False`` (mpc.py:237: hard state box, no slack variables) is honoured on the device.
There is NO CPU fallback: if the HIP library or a GPU is missing, ``setup()`` raises.
(Tests that exercise only the host logic inject their own ``prob`` object.)
"""
import warnings

import numpy as np
import scipy.sparse as sparse

from . import qp_build

# (attribute, expected length attribute, ravel?, message) -- mirrors the checks of mpc.py:107-223
_ERR = {
    'x0': "x0 should be an array of dimension (nx,)!",
    'xref': "xref should be either a vector of shape (nx,) or a matrix of shape (Np+1, nx)!",
    'uref': "uref should be a vector of shape (nu,)!",
    'uminus1': "uminus1 should be a vector of shape (nu,)!",
    'Qx': "Qx should be a matrix of shape (nx, nx)!",
    'QxN': "QxN should be a square matrix of shape (nx, nx)!",
    'Qu': "Qu should be a square matrix of shape (nu, nu)!",
    'QDu': "QDu should be a square matrix of shape (nu, nu)!",
    'xmin': "xmin should be a vector of shape (nx,)!",
    'xmax': "xmax should be a vector of shape (nx,)!",
    'umin': "umin should be a vector of shape (nu,)!",
    'umax': "umax should be a vector of shape (nu,)!",
    'Dumin': "Dumin should be a vector of shape (nu,)!",
    'Dumax': "Dumax should be a vector of shape (nu,)!",
}


def _vector_like(v):
    """The reference's ``__is_vector__`` (mpc.py:8-17): 1-D, or 2-D with a leading 1 (or an
    empty second axis); anything else -- including (k, 1) columns -- is rejected."""
    if v.ndim == 1:
        return True
    return v.ndim == 2 and (v.shape[0] == 1 or v.shape[1] == 0)


def _matrix_like(M):
    return M.ndim == 2


class MPCController:
    """Linear constrained MPC controller; see the reference class docstring (mpc.py:28-74)
    for the meaning of every argument.  All arguments, defaults and failure modes match."""

    def __init__(self, Ad, Bd, Np=20, Nc=None,
                 x0=None, xref=None, uref=None, uminus1=None,
                 Qx=None, QxN=None, Qu=None, QDu=None,
                 xmin=None, xmax=None, umin=None, umax=None, Dumin=None, Dumax=None,
                 eps_feas=1e6, eps_rel=1e-3, eps_abs=1e-3):
        if not (_matrix_like(Ad) and Ad.shape[0] == Ad.shape[1]):
            raise ValueError("Ad should be a square matrix of dimension (nx,nx)!")
        self.Ad = Ad
        self.nx = nx = Ad.shape[0]
        if not (_matrix_like(Bd) and Bd.shape[0] == nx):
            raise ValueError("Bd should be a matrix of dimension (nx, nu)!")
        self.Bd = Bd
        self.nu = nu = Bd.shape[1]
        if not Np > 1:
            raise ValueError("Np should be > 1!")
        self.Np = Np
        if Nc is not None and not Nc <= Np:
            raise ValueError("Nc should be <= Np!")
        self.Nc = Np if Nc is None else Nc

        def vec(name, val, size, default, ravel):
            if val is None:
                return default()
            if not (_vector_like(val) and val.size == size):
                raise ValueError(_ERR[name])
            return val.ravel() if ravel else val

        def mat(name, val, size, default):
            if val is None:
                return default()
            if not (_matrix_like(val) and val.shape[0] == size and val.shape[1] == size):
                raise ValueError(_ERR[name])
            return val

        self.x0 = vec('x0', x0, nx, lambda: np.zeros(nx), True)
        if xref is None:
            self.xref = np.zeros(nx)
        elif _vector_like(xref) and xref.size == nx:
            self.xref = xref.ravel()
        elif _matrix_like(xref) and xref.shape[1] == nx and xref.shape[0] >= Np:
            self.xref = xref
        else:
            raise ValueError(_ERR['xref'])
        self.uref = vec('uref', uref, nu, lambda: np.zeros(nu), True)
        # NB: default uminus1 ALIASES uref, like mpc.py:141
        self.uminus1 = vec('uminus1', uminus1, nu, lambda: self.uref, False)

        self.Qx = mat('Qx', Qx, nx, lambda: np.zeros((nx, nx)))
        if QxN is None:
            self.QxN = self.Qx
        else:
            # the reference reads Qx.shape[1] in this check (mpc.py:153); with Qx=None that is an
            # AttributeError there as well
            if not (_matrix_like(QxN) and QxN.shape[0] == nx and Qx.shape[1] == nx):
                raise ValueError(_ERR['QxN'])
            self.QxN = QxN
        self.Qu = mat('Qu', Qu, nu, lambda: np.zeros((nu, nu)))
        self.QDu = mat('QDu', QDu, nu, lambda: np.zeros((nu, nu)))

        inf = np.inf
        self.xmin = vec('xmin', xmin, nx, lambda: -np.ones(nx) * inf, True)
        self.xmax = vec('xmax', xmax, nx, lambda: np.ones(nx) * inf, False)
        self.umin = vec('umin', umin, nu, lambda: -np.ones(nu) * inf, False)
        self.umax = vec('umax', umax, nu, lambda: np.ones(nu) * inf, False)
        self.Dumin = vec('Dumin', Dumin, nu, lambda: -np.ones(nu) * inf, False)
        self.Dumax = vec('Dumax', Dumax, nu, lambda: np.ones(nu) * inf, False)

        self.eps_feas = eps_feas
        self.Qeps = eps_feas * sparse.eye(nx)
        self.eps_rel = eps_rel
        self.eps_abs = eps_abs
        self.u_failure = self.uref      # returned by output() when the solve failed

        # hidden debug switches of mpc.py:233-238
        self.raise_error = False
        self.JX_ON = True
        self.JU_ON = True
        self.JDU_ON = True
        self.SOFT_ON = True
        self.COMPUTE_J_CNST = False

        # Solver instance.  Created lazily in setup() so that constructing a controller never
        # touches the GPU; tests may assign their own object here before setup().
        self.prob = None
        # extra solver settings forwarded to the device problem (additive surface)
        self.solver_settings = {}

        self.res = None
        self.P = None
        self._q = self._l = self._u = self._J_CNST = None
        self._stale = False           # q, l, u, J_CNST not yet refreshed after the last update() (device solver only)
        self.A = None
        self.P_X = None
        self.x0_rh = None
        self.uminus1_rh = None

    # ------------------------------------------------------------------------------------
    def _model_data(self):
        """Raw controller data handed to the device, which builds P,q,A,l,u itself."""
        d = lambda M: M.toarray() if sparse.issparse(M) else np.asarray(M, dtype=float)
        return dict(
            nx=self.nx, nu=self.nu, Np=self.Np, Nc=self.Nc,
            Ad=d(self.Ad), Bd=d(self.Bd),
            Qx=d(self.Qx) if self.JX_ON else np.zeros((self.nx, self.nx)),
            QxN=d(self.QxN) if self.JX_ON else np.zeros((self.nx, self.nx)),
            Qu=d(self.Qu) if self.JU_ON else np.zeros((self.nu, self.nu)),
            QDu=d(self.QDu) if self.JDU_ON else np.zeros((self.nu, self.nu)),
            xmin=self.xmin, xmax=self.xmax, umin=self.umin, umax=self.umax,
            Dumin=self.Dumin, Dumax=self.Dumax, eps_feas=float(self.eps_feas),
            x0=np.asarray(self.x0_rh, dtype=float), uminus1=np.asarray(self.uminus1_rh, dtype=float),
            xref=np.asarray(self.xref, dtype=float), uref=np.asarray(self.uref, dtype=float),
            SOFT_ON=bool(self.SOFT_ON),
        )

    def _step_data(self):
        return dict(x0=np.asarray(self.x0_rh, dtype=float),
                    uminus1=np.asarray(self.uminus1_rh, dtype=float),
                    xref=np.asarray(self.xref, dtype=float))

    def setup(self, solve=True):
        """Build the QP and set the solver up (mpc.py:254-269); optionally solve it."""
        self.x0_rh = np.copy(self.x0)
        self.uminus1_rh = np.copy(self.uminus1)
        self._compute_QP_matrices_()
        if self.prob is None:
            from .solver import DeviceProblem   # raises loudly without the HIP library / a GPU
            self.prob = DeviceProblem()
        # the reference passes eps_abs=self.eps_rel, eps_rel=self.eps_abs (swapped, mpc.py:266)
        self.prob.setup(self.P, self.q, self.A, self.l, self.u, warm_start=True, verbose=False,
                        eps_abs=self.eps_rel, eps_rel=self.eps_abs,
                        mpc=self._model_data(), **self.solver_settings)
        if solve:
            self.solve()

    def output(self, return_x_seq=False, return_u_seq=False, return_eps_seq=False,
               return_status=False, return_obj_val=False):
        """First optimal input (or ``u_failure``) and optional info dict (mpc.py:271-336)."""
        Np, Nc, nx, nu = self.Np, self.Nc, self.nx, self.nu
        ou = (Np + 1) * nx
        oe = ou + Nc * nu
        if self.res.info.status == 'solved':
            uMPC = self.res.x[ou:ou + nu]
        else:
            uMPC = self.u_failure
        info = {}
        if return_x_seq:
            info['x_seq'] = self.res.x[0:ou].reshape(-1, nx)
        if return_u_seq:
            info['u_seq'] = self.res.x[ou:oe].reshape(-1, nu)
        if return_eps_seq:
            info['eps_seq'] = self.res.x[oe:oe + ou].reshape(-1, nx)
        if return_status:
            info['status'] = self.res.info.status
        if return_obj_val:
            info['obj_val'] = self.res.info.obj_val + self.J_CNST
        self.uminus1_rh = uMPC
        return uMPC if len(info) == 0 else (uMPC, info)

    def update(self, x, u=None, xref=None, solve=True):
        """New measurement (and optionally u_{-1}, xref): refresh q,l,u and re-solve (mpc.py:338-364)."""
        self.x0_rh = x
        if u is not None:
            self.uminus1_rh = u
        if xref is not None:
            self.xref = xref
        self._update_QP_matrices_()
        if solve:
            self.solve()
        elif hasattr(self.prob, 'flush'):
            self.prob.flush()                 # (the device solver sends the step data with the solve; without one, now)

    def solve(self):
        """Warm-started solve (mpc.py:366-375)."""
        self.res = self.prob.solve()
        if self.res.info.status != 'solved':
            warnings.warn('OSQP did not solve the problem!')
            if self.raise_error:
                raise ValueError('OSQP did not solve the problem!')

    def __controller_function__(self, x, u, xref=None):
        """u = K(x, u_{-1}) as a pure function (mpc.py:377-384)."""
        self.update(x, u, xref=xref, solve=True)
        return self.output()

    def unconstrained_gains(self, tol=1e-13):
        """Gain matrices of this controller's law without inequality constraints, U* = K_x0 x0 + K_xref xref + K_uref uref + K_um1 u_{-1}
        (test_scripts/alternative/unconstrained.py:170-183), computed on the device (pympc_amd/unconstrained.py: one KKT factorization, a
        few batched solves).  An addition to the reference's class; output() does not use it."""
        from .unconstrained import unconstrained_gains, GainSolver
        d = lambda M: np.asarray(M.toarray() if hasattr(M, 'toarray') else M, dtype=float)        # (scipy.sparse inputs are accepted like in the reference)
        gs = getattr(self, '_gain_solver', None)
        if gs is None or gs.tol != tol:
            gs = self._gain_solver = GainSolver(self.nx, self.nu, self.Np, self.Nc, tol=tol)      # (kept: a second call allocates nothing)
        return unconstrained_gains(d(self.Ad), d(self.Bd), self.Np, self.Nc, Qx=d(self.Qx), QxN=d(self.QxN), Qu=d(self.Qu), QDu=d(self.QDu), tol=tol, solver=gs)

    # ------------------------------------------------------------------------------------
    # q, l, u, J_CNST are public attributes of the reference (mpc.py:598-606).  The device solver rebuilds them itself from
    # (x0, u_{-1}, xref), so after update() the host copies are refreshed only when somebody reads them.
    def _fresh(self):
        if self._stale:
            self._stale = False
            # the QP that was handed to the solver: built from the values update() saw, not from what x0_rh /
            # uminus1_rh / xref have become since (output() moves uminus1_rh on, callers may reuse their x array)
            live = self.x0_rh, self.uminus1_rh, self.xref
            self.x0_rh, self.uminus1_rh, self.xref = self._snap
            try:
                self._q, self._J_CNST = qp_build.refresh_vectors(self)
            finally:
                self.x0_rh, self.uminus1_rh, self.xref = live

    q = property(lambda self: (self._fresh(), self._q)[1], lambda self, v: setattr(self, '_q', v))
    l = property(lambda self: (self._fresh(), self._l)[1], lambda self, v: setattr(self, '_l', v))
    u = property(lambda self: (self._fresh(), self._u)[1], lambda self, v: setattr(self, '_u', v))
    J_CNST = property(lambda self: (self._fresh(), self._J_CNST)[1], lambda self, v: setattr(self, '_J_CNST', v))

    def _update_QP_matrices_(self):
        from .solver import DeviceProblem
        if isinstance(self.prob, DeviceProblem):
            self._stale = True                    # refreshed on first read, from a snapshot of this call's inputs
            self._snap = (np.array(self.x0_rh, dtype=float), np.array(self.uminus1_rh, dtype=float), np.array(self.xref, dtype=float))
            self.prob.update(mpc_step=self._snap)      # (the snapshot itself: private copies, float64, contiguous -- nothing to convert again on the way down)
        else:                                     # a solver that wants the vectors (the oracle in the tests)
            self._stale = False
            self.q, self.J_CNST = qp_build.refresh_vectors(self)
            self.prob.update(l=self.l, u=self.u, q=self.q, mpc_step=self._step_data())

    def _compute_QP_matrices_(self):
        self.P, self.q, self.A, self.l, self.u, self.P_X, self.J_CNST = qp_build.build_qp(self)
