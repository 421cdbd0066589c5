"""Python face of the HIP solver.

``BatchProblem``  -- a batch of independent MPC instances of equal dimensions living on one GPU
                     (thin, 1:1 wrapper over the C ABI of include/mpcqp.h).
``DeviceProblem`` -- the object ``MPCController`` keeps in ``self.prob`` where the reference keeps
                     ``osqp.OSQP()`` (pyMPC/mpc.py:241): ``setup / update / solve`` with the call
                     shapes of mpc.py:266, 454, 369, returning an osqp-like result object.

Arrays may be numpy ndarrays (host) or torch CUDA tensors (device-resident); both are passed as
raw pointers, PyTorch being used only as a device-memory container.
"""
import ctypes as C
import math

import numpy as np

from . import _lib

_SETTING_NAMES = [f[0] for f in _lib.Settings._fields_]
# accepted for call compatibility with osqp, without effect on this solver
_IGNORED_SETTINGS = ('verbose', 'polish', 'linsys_solver', 'time_limit', 'scaled_termination', 'delta',
                     'polish_refine_iter', 'adaptive_rho_fraction')


# fixed when the handle is created / set up: mpcqp_update_settings keeps the handle's own values (include/mpcqp.h)
_FIXED_AT_CREATE = ('rho', 'sigma', 'scaling', 'soft_constraints', 'backend', 'tuning')

_STATUS_STRINGS = {}       # status code -> OSQP's string (filled from mpcqp_status_string on first use)


def _ptr(a):
    if a is None:
        return None
    if hasattr(a, 'data_ptr'):             # torch tensor (device or host)
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(a.ctypes.data)


def _prep(a, shape, name):
    """float64, C-contiguous, exact shape; returns an object that keeps the memory alive."""
    if a is None:
        raise ValueError('%s is required' % name)
    if hasattr(a, 'data_ptr'):
        import torch
        if a.numel() != int(np.prod(shape)):
            a = a.expand(shape)
        if a.dtype != torch.float64 or not a.is_contiguous() or tuple(a.shape) != tuple(shape):
            a = a.to(torch.float64).reshape(shape).contiguous()
        return a
    a = np.asarray(a, dtype=np.float64)
    if a.size == int(np.prod(shape)):
        return np.ascontiguousarray(a.reshape(shape))
    return np.ascontiguousarray(np.broadcast_to(a, shape))


BACKENDS = dict(auto=_lib.BACKEND_AUTO, sweeps=_lib.BACKEND_SWEEPS, dense=_lib.BACKEND_DENSE, bcr=_lib.BACKEND_BCR, bcr8=_lib.BACKEND_BCR8, bcrt=_lib.BACKEND_BCRT)
_forced = {}               # settings every handle made inside a ``forced_settings`` block gets (tests: one parity suite per KKT backend)


class forced_settings:
    """``with forced_settings(backend='bcr', tuning=TUNE_NO_BALANCE): ...`` -- every problem created inside the block gets these
    ``mpcqp_settings`` fields on top of what its caller asked for.  This is how the test-suite forces a KKT backend
    (``mpcqp_settings.backend``) through code that builds its own controllers; the library itself reads no environment variable."""

    def __init__(self, **kw):
        if isinstance(kw.get('backend'), str):
            kw['backend'] = BACKENDS[kw['backend']]
        self.kw = kw

    def __enter__(self):
        self.old = dict(_forced)
        _forced.update(self.kw)
        return self

    def __exit__(self, *exc):
        _forced.clear()
        _forced.update(self.old)
        return False


def make_settings(**kw):
    L = _lib.load()
    s = _lib.Settings()
    L.mpcqp_default_settings(C.byref(s))
    kw = dict(kw, **_forced)
    if isinstance(kw.get('backend'), str):
        kw['backend'] = BACKENDS[kw['backend']]
    for k, v in kw.items():
        if k in _SETTING_NAMES:
            setattr(s, k, v)
        elif k in _IGNORED_SETTINGS:
            if k == 'scaled_termination' and v:
                raise NotImplementedError('scaled_termination=True is not implemented')
        else:
            raise TypeError('unknown solver setting %r' % k)
    return s


class Result:
    """Mimics osqp's results object as far as pyMPC reads it (mpc.py:301-327)."""

    class _Info:
        pass

    def __init__(self, x, y, info, status_str):
        self.x = x
        self.y = y
        self.info = Result._Info()
        self.info.status_val = int(info.status)
        self.info.status = status_str
        self.info.iter = int(info.iter)
        self.info.rho_updates = int(info.rho_updates)
        self.info.obj_val = float(info.obj_val)
        self.info.pri_res = float(info.pri_res)
        self.info.dua_res = float(info.dua_res)
        self.info.rho_estimate = float(info.rho)


class BatchProblem:
    """``batch`` independent MPC problems (nx, nu, Np, Nc) on GPU ``device``."""

    def __init__(self, batch, nx, nu, Np, Nc=None, device=0, stream=None, **settings):
        self._L = _lib.load()
        self._h = C.c_void_p()
        if self._L.mpcqp_device_count() <= 0:
            raise RuntimeError('pympc_amd needs an AMD GPU (no HIP device visible); there is no CPU fallback')
        self.batch, self.nx, self.nu, self.Np = int(batch), int(nx), int(nu), int(Np)
        self.Nc = int(Np if Nc is None else Nc)
        self.settings = make_settings(**settings)
        rc = self._L.mpcqp_create(C.byref(self._h), int(device), self.batch, self.nx, self.nu, self.Np, self.Nc,
                                  C.byref(self.settings))
        if rc == -4:
            raise NotImplementedError(self._L.mpcqp_last_error().decode())
        _lib.check(rc, 'mpcqp_create')
        self._finish_init(stream)

    @classmethod
    def from_matrices(cls, P, A, batch=1, device=0, stream=None, nx=None, nu=None, **settings):
        """The seam of mpc.py:266 with the matrices themselves (mpcqp_create_csc): the LIBRARY reads nx, nu, Np, Nc out of
        the sparsity patterns of P and A (scipy sparse, shared by the batch) and later, in ``setup_csc``, the controller
        data out of their values -- refusing anything that is not pyMPC's QP (``NotAnMPCQP``)."""
        import scipy.sparse as sp
        from .qp_recover import NotAnMPCQP
        self = cls.__new__(cls)
        self._L = _lib.load()
        self._h = C.c_void_p()
        self.batch = int(batch)
        self.settings = make_settings(**settings)
        Pc, Ac = sp.csc_matrix(P), sp.csc_matrix(A)
        Pc.sort_indices(); Ac.sort_indices()
        self._csc = (Pc, Ac)
        pp, pi = np.ascontiguousarray(Pc.indptr, dtype=np.int64), np.ascontiguousarray(Pc.indices, dtype=np.int32)
        ap, ai = np.ascontiguousarray(Ac.indptr, dtype=np.int64), np.ascontiguousarray(Ac.indices, dtype=np.int32)
        rc = self._L.mpcqp_create_csc(C.byref(self._h), int(device), self.batch, int(Pc.shape[0]), int(Ac.shape[0]), _ptr(pp), _ptr(pi), _ptr(ap), _ptr(ai),
                                      int(nx or 0), int(nu or 0), C.byref(self.settings))
        if rc == -4:        # MPCQP_ERR_UNSUPPORTED: either not pyMPC's QP, or a valid one beyond what the device path implements
            msg = self._L.mpcqp_last_error().decode()
            if msg.startswith('mpcqp_create_csc: not an MPC QP'):
                raise NotAnMPCQP(msg)
            raise NotImplementedError(msg)
        if rc == -3:
            raise RuntimeError('pympc_amd needs an AMD GPU (no HIP device visible); there is no CPU fallback')
        _lib.check(rc, 'mpcqp_create_csc')
        d = [C.c_int() for _ in range(4)]
        _lib.check(self._L.mpcqp_get_shape(self._h, *[C.byref(v) for v in d]), 'mpcqp_get_shape')
        self.nx, self.nu, self.Np, self.Nc = (v.value for v in d)
        self._finish_init(stream)
        return self

    def setup_csc(self, P_val, A_val, q, l, u):
        """Values of P [B, nnz(P)] and A [B, nnz(A)] in the index order given to ``from_matrices`` (sorted CSC), q [B, n], l, u [B, m]
        (host arrays): recovered, verified by a rebuild and set up by the library (mpcqp_setup_csc)."""
        from .qp_recover import NotAnMPCQP
        Pc, Ac = self._csc
        pv = _prep(P_val, (self.batch, Pc.nnz), 'P_val'); av = _prep(A_val, (self.batch, Ac.nnz), 'A_val')
        qa, la, ua = _prep(q, (self.batch, self.n), 'q'), _prep(l, (self.batch, self.m), 'l'), _prep(u, (self.batch, self.m), 'u')
        rc = self._L.mpcqp_setup_csc(self._h, _ptr(pv), _ptr(av), _ptr(qa), _ptr(la), _ptr(ua))
        if rc == -4:
            raise NotAnMPCQP(self._L.mpcqp_last_error().decode())
        _lib.check(rc, 'mpcqp_setup_csc')

    _hin = None                # (step_host of a single controller: input buffer, made on first use)

    def _finish_init(self, stream):
        n, m, fd, nnzL = C.c_int(), C.c_int(), C.c_int64(), C.c_int64()
        _lib.check(self._L.mpcqp_get_dims(self._h, C.byref(n), C.byref(m), C.byref(fd), C.byref(nnzL)), 'mpcqp_get_dims')
        self.n, self.m, self.factor_doubles, self.nnzL = n.value, m.value, fd.value, nnzL.value
        if stream is not None:
            _lib.check(self._L.mpcqp_set_stream(self._h, C.c_void_p(int(stream))), 'mpcqp_set_stream')
        self._keep = []

    def close(self):
        if getattr(self, '_h', None) and self._h.value:
            self._L.mpcqp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference-shaped calls ---------------------------------------------------------------
    def setup(self, Ad, Bd, Qx, QxN, Qu, QDu, xmin, xmax, umin, umax, Dumin, Dumax, uref, eps_feas,
              x0, uminus1, xref):
        B, nx, nu = self.batch, self.nx, self.nu
        xref_rows = self._xref_rows(xref)
        arrs = dict(
            Ad=_prep(Ad, (B, nx, nx), 'Ad'), Bd=_prep(Bd, (B, nx, nu), 'Bd'),
            Qx=_prep(Qx, (B, nx, nx), 'Qx'), QxN=_prep(QxN, (B, nx, nx), 'QxN'),
            Qu=_prep(Qu, (B, nu, nu), 'Qu'), QDu=_prep(QDu, (B, nu, nu), 'QDu'),
            xmin=_prep(xmin, (B, nx), 'xmin'), xmax=_prep(xmax, (B, nx), 'xmax'),
            umin=_prep(umin, (B, nu), 'umin'), umax=_prep(umax, (B, nu), 'umax'),
            Dumin=_prep(Dumin, (B, nu), 'Dumin'), Dumax=_prep(Dumax, (B, nu), 'Dumax'),
            uref=_prep(uref, (B, nu), 'uref'), eps_feas=_prep(eps_feas, (B, 1), 'eps_feas'))
        model = _lib.Model()
        for k, v in arrs.items():
            setattr(model, k, C.cast(_ptr(v), C.POINTER(C.c_double)))
        x0a = _prep(x0, (B, nx), 'x0')
        uma = _prep(uminus1, (B, nu), 'uminus1')
        xra = _prep(xref, (B, xref_rows * nx), 'xref')
        self._keep = [arrs, x0a, uma, xra]
        _lib.check(self._L.mpcqp_setup(self._h, C.byref(model), _ptr(x0a), _ptr(uma), _ptr(xra), xref_rows), 'mpcqp_setup')
        _lib.check(self._L.mpcqp_synchronize(self._h), 'mpcqp_synchronize')
        self._keep = []

    def setup_qp(self, Ad, Bd, Qx, QxN, Qu, QDu, eps_feas, q, l, u, uref=None):
        """The solver seam of mpc.py:266 with caller-built vectors (mpcqp_setup_qp): the matrices enter through the
        blocks they are made of (pympc_amd.qp_recover reads them out of a reference-layout P, A), q [B,n], l, u [B,m]
        verbatim."""
        B, nx, nu = self.batch, self.nx, self.nu
        arrs = dict(Ad=_prep(Ad, (B, nx, nx), 'Ad'), Bd=_prep(Bd, (B, nx, nu), 'Bd'), Qx=_prep(Qx, (B, nx, nx), 'Qx'),
                    QxN=_prep(QxN, (B, nx, nx), 'QxN'), Qu=_prep(Qu, (B, nu, nu), 'Qu'), QDu=_prep(QDu, (B, nu, nu), 'QDu'),
                    eps_feas=_prep(eps_feas, (B, 1), 'eps_feas'))
        if uref is not None:
            arrs['uref'] = _prep(uref, (B, nu), 'uref')
        model = _lib.Model()
        for k, v in arrs.items():
            setattr(model, k, C.cast(_ptr(v), C.POINTER(C.c_double)))
        qa, la, ua = _prep(q, (B, self.n), 'q'), self._bound(l, 'l'), self._bound(u, 'u')
        _lib.check(self._L.mpcqp_setup_qp(self._h, C.byref(model), _ptr(qa), _ptr(la), _ptr(ua)), 'mpcqp_setup_qp')
        _lib.check(self._L.mpcqp_synchronize(self._h), 'mpcqp_synchronize')

    def _bound(self, v, name):
        return None if v is None else _prep(v, (self.batch, self.m), name)

    def update_vectors(self, q=None, l=None, u=None):
        """osqp's update(q=, l=, u=) (mpc.py:454) with caller-built vectors.  q may be None (unchanged); l and u go together --
        both or neither: the equality rows l[:nx] == u[:nx] carry x0 (the library refuses one without the other)."""
        if (l is None) != (u is None):
            raise ValueError('update_vectors: give l and u together (the equality rows l[:nx] == u[:nx] carry x0)')
        qa = None if q is None else _prep(q, (self.batch, self.n), 'q')
        la, ua = self._bound(l, 'l'), self._bound(u, 'u')
        _lib.check(self._L.mpcqp_update_vectors(self._h, _ptr(qa), _ptr(la), _ptr(ua)), 'mpcqp_update_vectors')

    def _xref_rows(self, xref):
        shape = tuple(xref.shape)
        per = math.prod(shape[1:]) if len(shape) > 1 and shape[0] == self.batch else math.prod(shape)
        if per == self.nx:
            return 1
        if per == (self.Np + 1) * self.nx:
            return self.Np + 1
        raise ValueError('xref must hold nx or (Np+1)*nx values per instance')

    def update(self, x0=None, uminus1=None, xref=None):
        B, nx, nu = self.batch, self.nx, self.nu
        rows = 1
        a = _prep(x0, (B, nx), 'x0') if x0 is not None else None
        b = _prep(uminus1, (B, nu), 'uminus1') if uminus1 is not None else None
        c = None
        if xref is not None:
            rows = self._xref_rows(xref)
            c = _prep(xref, (B, rows * nx), 'xref')
        self._keep = [a, b, c]
        _lib.check(self._L.mpcqp_update(self._h, _ptr(a), _ptr(b), _ptr(c), rows), 'mpcqp_update')
        if any(hasattr(v, 'ctypes') for v in self._keep if v is not None):
            # pageable host memory: the async copy has completed or been staged when the call returns
            pass

    def update_settings(self, **kw):
        for k, v in kw.items():
            if k not in _SETTING_NAMES:
                raise TypeError('unknown solver setting %r' % k)
            if k in _FIXED_AT_CREATE and v != getattr(self.settings, k):
                # (mpcqp_update_settings keeps the handle's own value of these: they shape the problem, the factorization or the kernel choice)
                raise ValueError('solver setting %r is fixed when the problem is created / set up; make a new problem to change it' % k)
            setattr(self.settings, k, v)
        _lib.check(self._L.mpcqp_update_settings(self._h, C.byref(self.settings)), 'mpcqp_update_settings')

    def warm_start(self, x=None, y=None):
        a = _prep(x, (self.batch, self.n), 'x') if x is not None else None
        b = _prep(y, (self.batch, self.m), 'y') if y is not None else None
        _lib.check(self._L.mpcqp_warm_start(self._h, _ptr(a), _ptr(b)), 'mpcqp_warm_start')
        _lib.check(self._L.mpcqp_synchronize(self._h), 'mpcqp_synchronize')

    def solve_async(self):
        _lib.check(self._L.mpcqp_solve(self._h), 'mpcqp_solve')

    def synchronize(self):
        _lib.check(self._L.mpcqp_synchronize(self._h), 'mpcqp_synchronize')

    def mpc_step(self, x0, uminus1=None, xref=None, out=None):
        """One control step ``u = K(x, u_{-1})`` (mpcqp_mpc_step = update + warm solve + output with the u_failure rule)."""
        B, nx, nu = self.batch, self.nx, self.nu
        a = _prep(x0, (B, nx), 'x0')
        b = _prep(uminus1, (B, nu), 'uminus1') if uminus1 is not None else None
        c, rows = None, 1
        if xref is not None:
            rows = self._xref_rows(xref)
            c = _prep(xref, (B, rows * nx), 'xref')
        if out is None:
            out = np.empty((B, nu))
        _lib.check(self._L.mpcqp_mpc_step(self._h, _ptr(a), _ptr(b), _ptr(c), rows, _ptr(out)), 'mpcqp_mpc_step')
        return out

    def step_host(self, x0=None, uminus1=None, xref=None):
        """update(x0, uminus1, xref) + warm-started solve + solution in ONE library call with host arrays (mpcqp_step_host):
        the latency path of a single controller.  Returns ``(x [B,n], y [B,m], info[B])`` like ``solution()``."""
        B, nx, nu = self.batch, self.nx, self.nu
        if B == 1 and x0 is not None and uminus1 is not None and xref is not None:
            return self._step_host_one(x0, uminus1, xref)
        f64 = lambda v, cols: v if (type(v) is np.ndarray and v.dtype == np.float64 and v.flags.c_contiguous and v.size == B * cols) \
            else np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(B, cols))      # (a conforming array goes down as it is: same memory, whatever its shape)
        a = None if x0 is None else f64(x0, nx)
        b = None if uminus1 is None else f64(uminus1, nu)
        c, rows = None, 1
        if xref is not None:
            rows = self._xref_rows(xref)
            c = f64(xref, rows * nx)
        x, y = np.empty((B, self.n)), np.empty((B, self.m))
        info = (_lib.Info * B)()
        _lib.check(self._L.mpcqp_step_host(self._h, _ptr(a), _ptr(b), _ptr(c), rows, _ptr(x), _ptr(y), C.cast(info, C.c_void_p)), 'mpcqp_step_host')
        return x, y, info

    def _step_host_one(self, x0, uminus1, xref):
        """step_host for ONE controller with the Python side trimmed (a step is ~ 60 us in all: every microsecond here shows): the inputs are
        copied into a buffer the problem keeps -- its three addresses are plain integers computed once -- and x, y come back as views of one
        ctypes block (``ndarray.ctypes.data`` costs ~ 1 us per array, ``np.prod`` of a shape more)."""
        nx, nu, n, m = self.nx, self.nu, self.n, self.m
        hb = self._hin
        if hb is None:
            hb = self._hin = np.empty(nx + nu + (self.Np + 1) * nx)
            p = hb.ctypes.data
            self._hin_p = (p, p + 8 * nx, p + 8 * (nx + nu))
            self._out_t = C.c_double * (n + m)
        xr = xref if type(xref) is np.ndarray else np.asarray(xref, dtype=np.float64)
        nxr = xr.size
        if nxr != nx and nxr != (self.Np + 1) * nx:
            raise ValueError('xref must hold nx or (Np+1)*nx values per instance')
        hb[:nx] = np.reshape(x0, nx); hb[nx:nx + nu] = np.reshape(uminus1, nu); hb[nx + nu:nx + nu + nxr] = xr.reshape(nxr)
        cb = self._out_t()
        info = (_lib.Info * 1)()
        a, b, c = self._hin_p
        _lib.check(self._L.mpcqp_step_host(self._h, a, b, c, nxr // nx, cb, C.addressof(cb) + 8 * n, info), 'mpcqp_step_host')
        buf = np.frombuffer(cb, dtype=np.float64)
        return buf[:n].reshape(1, n), buf[n:].reshape(1, m), info

    def mpc_run(self, nsteps, w=None, Ap=None, Bp=None, out=None, xref_traj=None, estimator=None):
        """Device-side receding-horizon loop (mpcqp_mpc_loop): ``nsteps`` closed-loop steps
        ``u = output(); x = Ap x + Bp u + w[k]; update(x)`` of every instance without host round trips.

        ``w`` [nsteps, batch, nx] (numpy or torch device tensor) or None; ``Ap``/``Bp`` default to the controller model;
        ``xref_traj`` [nsteps, batch, rows*nx] gives the reference of the solve after step k (rows as last uploaded).
        ``estimator = dict(C=[B,ny,nx], L=[B,nx,ny], x_true=[B,nx], v=[nsteps,B,ny] or None)`` switches output feedback on
        (pyMPC/kalman.py:109-134): the controller is updated with the estimate, ``x_true`` is advanced in place.
        Returns ``(x_traj [nsteps+1,B,nx], u_traj [nsteps,B,nu], status [nsteps,B] int32, iters [nsteps,B] int32)`` as
        numpy arrays (plus ``xhat_traj, y_traj`` with an estimator), or fills the arrays/tensors given in ``out``."""
        K, B, nx, nu = int(nsteps), self.batch, self.nx, self.nu
        io = _lib.Loop()
        keep = []

        def inp(a, shape, name):
            if a is None:
                return None
            a = _prep(a, shape, name); keep.append(a)
            return _ptr(a)

        io.w = inp(w, (K, B, nx), 'w'); io.Ap = inp(Ap, (B, nx, nx), 'Ap'); io.Bp = inp(Bp, (B, nx, nu), 'Bp')
        if xref_traj is not None:
            shp = tuple(xref_traj.shape)
            if B == 1 and len(shp) == 2 and shp[0] == K:          # a single controller's [nsteps, nx] reference sequence
                xref_traj = xref_traj.reshape(K, 1, shp[1]); shp = tuple(xref_traj.shape)
            if len(shp) < 3 or shp[0] != K or shp[1] != B:
                raise ValueError('xref_traj must be [nsteps, batch, nx] or [nsteps, batch, Np+1, nx] (or flattened [nsteps, batch, (Np+1)*nx])')
            per = int(np.prod(shp[2:]))
            if per not in (nx, (self.Np + 1) * nx):
                raise ValueError('xref_traj must hold nx or (Np+1)*nx values per step and instance')
            io.xref_rows = per // nx          # the library checks it again and re-strides its copy accordingly
            io.xref_traj = inp(xref_traj, (K, B, per), 'xref_traj')
        ny = 0
        if estimator is not None:
            Cm = estimator['C']
            ny = int(tuple(Cm.shape)[-2])
            io.ny = ny
            io.C = inp(Cm, (B, ny, nx), 'C'); io.Lgain = inp(estimator['L'], (B, nx, ny), 'L')
            io.v = inp(estimator.get('v'), (K, B, ny), 'v')
            xt = estimator['x_true']
            if hasattr(xt, 'ctypes') and not (xt.dtype == np.float64 and xt.flags['C_CONTIGUOUS'] and xt.shape == (B, nx)):
                raise ValueError('estimator["x_true"] must be a C-contiguous float64 [batch, nx] array (it is updated in place)')
            keep.append(xt); io.x_true = _ptr(xt)
        if out is None:
            out = [np.empty((K + 1, B, nx)), np.empty((K, B, nu)), np.empty((K, B), dtype=np.int32), np.empty((K, B), dtype=np.int32)]
            if ny:
                out += [np.empty((K + 1, B, nx)), np.empty((K, B, ny))]
        io.x_traj, io.u_traj, io.status_traj, io.iter_traj = (_ptr(o) for o in out[:4])
        if ny and len(out) >= 6:
            io.xhat_traj, io.y_traj = _ptr(out[4]), _ptr(out[5])
        _lib.check(self._L.mpcqp_mpc_loop(self._h, K, C.byref(io)), 'mpcqp_mpc_loop')
        return tuple(out)

    def solution(self, want_y=True):
        x = np.empty((self.batch, self.n))
        y = np.empty((self.batch, self.m)) if want_y else None
        info = (_lib.Info * self.batch)()
        _lib.check(self._L.mpcqp_get_solution(self._h, _ptr(x), _ptr(y), C.cast(info, C.c_void_p)), 'mpcqp_get_solution')
        return x, y, info

    def infos(self):
        info = (_lib.Info * self.batch)()
        _lib.check(self._L.mpcqp_get_solution(self._h, None, None, C.cast(info, C.c_void_p)), 'mpcqp_get_solution')
        return info

    def u0(self, out=None):
        """First optimal input of every instance, [batch, nu] (numpy, or written into a torch tensor)."""
        if out is None:
            out = np.empty((self.batch, self.nu))
        _lib.check(self._L.mpcqp_get_u0(self._h, _ptr(out)), 'mpcqp_get_u0')
        return out

    def stats(self, reset=False):
        """Cumulative (iterations, residual evaluations, refactorizations, instance-solves) over the batch."""
        out = (C.c_uint64 * 4)()
        _lib.check(self._L.mpcqp_get_stats(self._h, out, int(bool(reset))), 'mpcqp_get_stats')
        return tuple(int(v) for v in out)

    def stream_bytes(self):
        """(bytes per ADMM iteration, per round, per solve) one instance streams by design (mpcqp_get_stream_bytes)."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(self._L.mpcqp_get_stream_bytes(self._h, C.byref(a), C.byref(b), C.byref(c)), 'mpcqp_get_stream_bytes')
        return a.value, b.value, c.value

    def mfma_per_iter(self):
        """Matrix-core instructions one instance issues per ADMM iteration (mpcqp_get_work)."""
        v = C.c_int64()
        _lib.check(self._L.mpcqp_get_work(self._h, C.byref(v)), 'mpcqp_get_work')
        return v.value

    def occupancy(self):
        """(workgroups of the solve kernel a compute unit holds, compute units of the device, threads per workgroup) -- mpcqp_get_occupancy."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self._L.mpcqp_get_occupancy(self._h, C.byref(a), C.byref(b), C.byref(c)), 'mpcqp_get_occupancy')
        return a.value, b.value, c.value

    def settings_dict(self):
        """The handle's solver settings as a dict (what `update_settings` last sent, on top of what the handle was created with)."""
        return {k: getattr(self.settings, k) for k in _SETTING_NAMES}

    def kernel_name(self, loop):
        buf = C.create_string_buffer(128)
        _lib.check(self._L.mpcqp_kernel_name(self._h, int(bool(loop)), buf, 128), 'mpcqp_kernel_name')
        return buf.value.decode()

    def launch_times(self, nsteps=0):
        """[batch, 2 + nsteps] uint64: entry and exit of every instance's workgroup in the last closed-loop launch and the end of each of its
        first ``nsteps`` (<= 64) steps, ticks of the GPU's 100 MHz clock."""
        out = np.zeros((self.batch, 2 + int(nsteps)), dtype=np.uint64)
        _lib.check(self._L.mpcqp_get_launch_times(self._h, _ptr(out), int(nsteps)), 'mpcqp_get_launch_times')
        return out

    def profile(self, enable=None, reset=False):
        """(milliseconds, launches) of the solve kernel k_mpc_run accumulated while profiling was enabled."""
        ms, nl = C.c_double(), C.c_int64()
        _lib.check(self._L.mpcqp_profile(self._h, -1 if enable is None else int(bool(enable)), C.byref(ms), C.byref(nl), int(bool(reset))), 'mpcqp_profile')
        return ms.value, nl.value

    def status_string(self, code):
        code = int(code)
        if code not in _STATUS_STRINGS:
            _STATUS_STRINGS[code] = self._L.mpcqp_status_string(code).decode()
        return _STATUS_STRINGS[code]

    # -- verification surface -----------------------------------------------------------------
    def export_qp(self):
        B, n, m = self.batch, self.n, self.m
        P, A = np.empty((B, n, n)), np.empty((B, m, n))
        q, l, u = np.empty((B, n)), np.empty((B, m)), np.empty((B, m))
        _lib.check(self._L.mpcqp_export_qp(self._h, _ptr(P), _ptr(A), _ptr(q), _ptr(l), _ptr(u)), 'mpcqp_export_qp')
        return P, q, A, l, u

    def scaling(self):
        D, E = np.empty((self.batch, self.n)), np.empty((self.batch, self.m))
        c, rho = np.empty(self.batch), np.empty(self.batch)
        _lib.check(self._L.mpcqp_get_scaling(self._h, _ptr(D), _ptr(E), _ptr(c), _ptr(rho)), 'mpcqp_get_scaling')
        return D, E, c, rho

    def kkt_solve(self, rhs):
        rhs = _prep(rhs, (self.batch, self.n), 'rhs')
        sol = np.empty((self.batch, self.n))
        _lib.check(self._L.mpcqp_debug_kkt_solve(self._h, _ptr(rhs), _ptr(sol)), 'mpcqp_debug_kkt_solve')
        return sol

    def iterate(self, iters):
        _lib.check(self._L.mpcqp_iterate(self._h, int(iters)), 'mpcqp_iterate')
        self.synchronize()

    def eq_solve(self, sweeps, cold=True, tol=0.0):
        """The equality-constrained part of the QP (dynamics rows only) by at most ``sweeps`` multiplier sweeps with the handle's KKT factor
        (mpcqp_eq_solve); an instance stops once a correction is below ``tol * max(1, |w|)``.  Returns [B, 5]: the four residual norms
        after the last sweep and the sweeps done; the solution is read with ``solution()``."""
        res = np.empty((self.batch, 5))
        _lib.check(self._L.mpcqp_eq_solve(self._h, int(sweeps), int(bool(cold)), float(tol), _ptr(res)), 'mpcqp_eq_solve')
        return res

    def share_factor(self):
        """One model, many states (test_scripts/example_mpc_function.py:105-111): every instance whose factorization inputs are bit-identical to
        instance 0's solves with ONE shared copy of its factor from now on (mpcqp_share_factor) -- call after ``setup`` with the same model and step
        data for every instance, then scatter the states with ``update``.  Results do not change.  Returns the number of instances sharing
        (0 for the register-resident backends)."""
        n = C.c_int(0)
        _lib.check(self._L.mpcqp_share_factor(self._h, C.byref(n)), 'mpcqp_share_factor')
        return int(n.value)

    def refactor(self):
        """Recompute every instance's KKT factor from its current rho (asynchronous; what one rho update costs)."""
        _lib.check(self._L.mpcqp_refactor(self._h), 'mpcqp_refactor')

    def iterate_state(self):
        x, z, y = np.empty((self.batch, self.n)), np.empty((self.batch, self.m)), np.empty((self.batch, self.m))
        _lib.check(self._L.mpcqp_get_iterate(self._h, _ptr(x), _ptr(z), _ptr(y)), 'mpcqp_get_iterate')
        return x, z, y


class DeviceProblem:
    """Single-instance adapter with osqp's call shapes, kept by ``MPCController.prob`` -- and usable in the REFERENCE's own
    class in place of ``osqp.OSQP()`` (mpc.py:241): ``setup(P, q, A, l, u, **settings)``, ``update(q=, l=, u=)``,
    ``solve()`` work on the caller's vectors (the seam of mpc.py:266,454,369).  ``pympc_amd.MPCController`` additionally
    passes ``mpc=`` / ``mpc_step=`` so that the device builds and refreshes q, l, u itself.

    ``nx, nu``: optional hints for reading the controller's dimensions out of P and A (they are inferred otherwise)."""

    def __init__(self, device=0, nx=None, nu=None):
        self.device = device
        self._bp = None
        self._hint = (nx, nu)
        self._model = None
        self._pending = None              # step data of an update() that has not reached the device yet (sent with the solve)

    def _setup_from_matrices(self, P, q, A, l, u, **settings):
        """setup(P, q, A, l, u) of mpc.py:266: the LIBRARY reads the controller out of the matrices and proves it by rebuilding them
        (mpcqp_create_csc / mpcqp_setup_csc, csrc/mpcqp_csc.h); ``NotAnMPCQP`` if they are not pyMPC's."""
        self._bp = BatchProblem.from_matrices(P, A, 1, device=self.device, nx=self._hint[0], nu=self._hint[1], **settings)
        Pc, Ac = self._bp._csc
        one = lambda v: np.asarray(v, dtype=float).reshape(1, -1)
        self._bp.setup_csc(Pc.data[None], Ac.data[None], one(q), one(l), one(u))
        self._model = dict(nx=self._bp.nx, nu=self._bp.nu, Np=self._bp.Np, Nc=self._bp.Nc)
        self.n, self.m = self._bp.n, self._bp.m
        self._model['SOFT_ON'] = self.n > (self._bp.Np + 1) * self._bp.nx + self._bp.Nc * self._bp.nu

    @staticmethod
    def _check_q(mdl, q):
        from . import qp_recover
        nq = (mdl['Np'] + 1) * mdl['nx'] + mdl['Nc'] * mdl['nu']
        q = np.asarray(q, dtype=float)
        if q.shape != (nq + ((mdl['Np'] + 1) * mdl['nx'] if mdl.get('SOFT_ON', True) else 0),) or np.any(q[nq:] != 0.0):
            raise qp_recover.NotAnMPCQP('q must have length n with a zero slack part (mpc.py:599)')

    def setup(self, P=None, q=None, A=None, l=None, u=None, mpc=None, **settings):
        if mpc is None:
            if P is None or q is None or A is None or l is None or u is None:
                raise ValueError('DeviceProblem.setup needs P, q, A, l, u (or the controller data, mpc=...)')
            return self._setup_from_matrices(P, q, A, l, u, **settings)
        xref = np.asarray(mpc['xref'], dtype=float)
        if xref.ndim == 2 and xref.shape[0] != mpc['Np'] + 1:
            raise ValueError('a time-varying xref must have exactly Np+1 rows')
        settings = dict(settings)
        settings.pop('soft_constraints', None)            # (SOFT_ON of the controller is the single source of truth)
        self._bp = BatchProblem(1, mpc['nx'], mpc['nu'], mpc['Np'], mpc['Nc'], device=self.device,
                                soft_constraints=int(bool(mpc.get('SOFT_ON', True))), **settings)
        one = lambda a: np.asarray(a, dtype=float)[None]
        self._bp.setup(one(mpc['Ad']), one(mpc['Bd']), one(mpc['Qx']), one(mpc['QxN']), one(mpc['Qu']), one(mpc['QDu']),
                       one(mpc['xmin']), one(mpc['xmax']), one(mpc['umin']), one(mpc['umax']),
                       one(mpc['Dumin']), one(mpc['Dumax']), one(mpc['uref']), np.array([[mpc['eps_feas']]]),
                       one(mpc['x0']), one(mpc['uminus1']), xref.reshape(1, -1))
        self.n, self.m = self._bp.n, self._bp.m

    def flush(self):
        """Send the step data of the last update() to the device now (it otherwise travels with the solve that follows)."""
        if self._pending is not None:
            pend, self._pending = self._pending, None
            self._bp.update(*pend)

    def update(self, q=None, l=None, u=None, mpc_step=None, **unsupported):
        if unsupported:
            raise NotImplementedError('DeviceProblem.update supports q, l, u (what pyMPC updates, mpc.py:454); got %s' % sorted(unsupported))
        if mpc_step is None:                              # the caller's vectors, verbatim
            self.flush()
            if self._model is None:
                from . import qp_recover
                raise qp_recover.NotAnMPCQP('update(q=, l=, u=) needs a problem set up from P, q, A, l, u')
            from . import qp_recover
            qp_recover.check_vectors(self._model, l, u)
            if q is not None:
                self._check_q(self._model, q)
            clip = lambda v: None if v is None else np.clip(np.asarray(v, dtype=float), -1e30, 1e30)[None]
            self._bp.update_vectors(None if q is None else np.asarray(q, dtype=float)[None], clip(l), clip(u))
            return
        # kept on the host until the solve that follows (mpc.py:338-364: update() = refresh + solve): one library call, one launch
        if isinstance(mpc_step, tuple):                   # (x0, uminus1, xref) as private float64 copies: MPCController's snapshot
            self._pending = mpc_step
        else:
            self._pending = (np.array(mpc_step['x0'], dtype=float).reshape(1, -1), np.array(mpc_step['uminus1'], dtype=float).reshape(1, -1),
                             np.array(mpc_step['xref'], dtype=float).reshape(1, -1))

    def update_settings(self, **kw):
        self._bp.update_settings(**kw)

    def warm_start(self, x=None, y=None):
        self.flush()
        self._bp.warm_start(None if x is None else np.asarray(x)[None], None if y is None else np.asarray(y)[None])

    def solve(self):
        if self._pending is not None:
            pend, self._pending = self._pending, None
            x, y, info = self._bp.step_host(*pend)
        else:
            self._bp.solve_async()
            x, y, info = self._bp.solution()
        return Result(x[0], y[0], info[0], self._bp.status_string(info[0].status))

    @property
    def batch_problem(self):
        self.flush()
        return self._bp
