"""ctypes front-end of oracle/libosqp_ref.so with the call surface the reference uses on `osqp`
(pyMPC/mpc.py:241,266,369,454): ``OSQP().setup(P,q,A,l,u,**settings)``, ``.update(q=,l=,u=)``,
``.solve()`` -> object with ``.x``, ``.y``, ``.info.status`` (OSQP's status strings),
``.info.obj_val``, ``.info.iter`` ...      TEST INFRASTRUCTURE, NOT PRODUCT CODE.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STATUS = {
    1: 'solved', 2: 'solved inaccurate', -2: 'maximum iterations reached',
    -3: 'primal infeasible', 3: 'primal infeasible inaccurate',
    -4: 'dual infeasible', 4: 'dual infeasible inaccurate',
    -7: 'problem non convex', -10: 'unsolved',
}


class Settings(C.Structure):
    _fields_ = [(k, C.c_double) for k in ('rho', 'sigma', 'alpha', 'eps_abs', 'eps_rel', 'eps_prim_inf',
                                          'eps_dual_inf', 'adaptive_rho_tolerance')] + \
               [(k, C.c_int) for k in ('max_iter', 'check_termination', 'scaling', 'adaptive_rho',
                                       'adaptive_rho_interval', 'warm_start', 'scaled_termination')]


class Info(C.Structure):
    _fields_ = [('status', C.c_int), ('iter', C.c_int), ('rho_updates', C.c_int),
                ('obj_val', C.c_double), ('pri_res', C.c_double), ('dua_res', C.c_double),
                ('rho_estimate', C.c_double)]


NATIVE = False        # oracle/cpu_bench.py sets this (before the first lib() call of its process) to time a build made
                      # with -O3 -march=native on the machine it runs on; everything else uses the portable build
BUILD_FLAGS = 'gcc -O3 -ffp-contract=off (portable build)'


def build(force=False):
    global BUILD_FLAGS
    src = os.path.join(_HERE, 'osqp_ref.c')
    if NATIVE:        # the copy oracle/cpu_bench.build_native() made on this machine, if it could
        so = os.path.join(_HERE, 'libosqp_ref_native.so')
        if os.path.exists(so):
            BUILD_FLAGS = 'gcc -O3 -march=native, built on this host'
            return so
    so = os.path.join(_HERE, 'libosqp_ref.so')
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'libosqp_ref.so'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        p64 = C.POINTER(C.c_int64)
        pd = C.POINTER(C.c_double)
        L.oracle_default_settings.argtypes = [C.POINTER(Settings)]
        L.oracle_setup.restype = C.c_void_p
        L.oracle_setup.argtypes = [C.c_int64, C.c_int64, p64, p64, pd, p64, p64, pd, pd, pd, pd, p64, C.POINTER(Settings)]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_update.argtypes = [C.c_void_p, pd, pd, pd]
        L.oracle_update.restype = C.c_int
        L.oracle_warm_start.argtypes = [C.c_void_p, pd, pd]
        L.oracle_solve.argtypes = [C.c_void_p, pd, pd, C.POINTER(Info)]
        L.oracle_iterate.argtypes = [C.c_void_p, C.c_int]
        L.oracle_get_scaling.argtypes = [C.c_void_p, pd, pd, pd]
        L.oracle_get_iterate.argtypes = [C.c_void_p, pd, pd, pd, pd]
        L.oracle_nnzL.argtypes = [C.c_void_p]
        L.oracle_mpc_closed_loop.restype = C.c_double
        L.oracle_mpc_closed_loop.argtypes = [C.c_void_p] + [C.c_int] * 5 + [pd] * 12 + [C.POINTER(C.c_int), pd, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        L.oracle_nnzL.restype = C.c_int64
        _LIB = L
    return _LIB


def _pd(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def kkt_ordering(P_triu, A, sigma=1e-6, rho=0.1):
    """Fill-reducing ordering of the KKT pattern (SuperLU's minimum-degree on A'+A)."""
    n, m = P_triu.shape[0], A.shape[0]
    Pf = P_triu + sp.triu(P_triu, 1).T
    K = sp.bmat([[Pf + sigma * sp.eye(n), A.T], [A, -1.0 / rho * sp.eye(m)]], format='csc')
    K.data = np.where(K.data == 0, 1e-300, K.data)
    lu = spla.splu(K, permc_spec='MMD_AT_PLUS_A', diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    return np.asarray(lu.perm_c, dtype=np.int64).argsort().astype(np.int64)  # perm[new] = old


class _Obj:
    pass


class OSQP:
    def __init__(self):
        self._w = None

    def __del__(self):
        try:
            if self._w:
                lib().oracle_free(self._w)
        except Exception:
            pass

    def setup(self, P, q, A, l, u, mpc=None, verbose=False, ordering='mmd', **settings):
        L = lib()
        s = Settings()
        L.oracle_default_settings(C.byref(s))
        for k, v in settings.items():
            if not hasattr(s, k):
                raise TypeError('unknown setting %r' % k)
            setattr(s, k, v)
        self.settings = s
        Pu = sp.triu(sp.csc_matrix(P), format='csc')
        Pu.sort_indices()
        Ac = sp.csc_matrix(A)
        Ac.sort_indices()
        self.n, self.m = Pu.shape[0], Ac.shape[0]
        perm = kkt_ordering(Pu, Ac, s.sigma, s.rho) if ordering == 'mmd' else None
        self._keep = [Pu.indptr.astype(np.int64), Pu.indices.astype(np.int64), Pu.data.astype(float),
                      Ac.indptr.astype(np.int64), Ac.indices.astype(np.int64), Ac.data.astype(float),
                      np.ascontiguousarray(q, dtype=float), np.ascontiguousarray(l, dtype=float),
                      np.ascontiguousarray(u, dtype=float)]
        k = self._keep
        self._w = L.oracle_setup(self.n, self.m, _p64(k[0]), _p64(k[1]), _pd(k[2]), _p64(k[3]), _p64(k[4]), _pd(k[5]),
                                 _pd(k[6]), _pd(k[7]), _pd(k[8]), _p64(perm) if perm is not None else None, C.byref(s))
        if not self._w:
            raise RuntimeError('oracle_setup failed (singular KKT?)')

    @property
    def nnzL(self):
        return lib().oracle_nnzL(self._w)

    def update(self, q=None, l=None, u=None, mpc_step=None):
        a = [None if v is None else np.ascontiguousarray(v, dtype=float) for v in (q, l, u)]
        if lib().oracle_update(self._w, _pd(a[0]), _pd(a[1]), _pd(a[2])):
            raise ValueError('lower bound must be lower than or equal to upper bound')

    def warm_start(self, x=None, y=None):
        a = [None if v is None else np.ascontiguousarray(v, dtype=float) for v in (x, y)]
        lib().oracle_warm_start(self._w, _pd(a[0]), _pd(a[1]))

    def closed_loop(self, K, x, noise):
        """`len(noise)` receding-horizon steps of controller `K` (whose solver this is; last solve in K.res) from plant state
        `x`, entirely in C (oracle_mpc_closed_loop): returns (seconds in update+solve, ADMM iterations, unsolved steps, x)."""
        nx, nu, Np, Nc = K.nx, K.nu, K.Np, K.Nc
        f = lambda a: np.ascontiguousarray(a, dtype=float)
        ou = (Np + 1) * nx
        q0 = f(K.q).copy()
        q0[ou:ou + nu] += f(K.QDu) @ f(K.uminus1_rh)                  # linear cost for u_{-1} = 0
        q, l, u = f(K.q).copy(), f(K.l).copy(), f(K.u).copy()
        xs, xsol = f(x).copy(), f(K.res.x).copy()
        status = C.c_int(K.res.info.status_val)
        noise = f(noise)
        it, bad = C.c_longlong(), C.c_longlong()
        arrs = [f(K.Ad), f(K.Bd), f(K.QDu), f(K.uref), f(np.clip(K.Dumin, -1e30, 1e30)), f(np.clip(K.Dumax, -1e30, 1e30)), q0, q, l, u, xs, xsol]
        t = lib().oracle_mpc_closed_loop(self._w, len(noise), nx, nu, Np, Nc, *[_pd(a) for a in arrs], C.byref(status), _pd(noise),
                                         C.byref(it), C.byref(bad))
        return t, it.value, bad.value, xs

    def iterate(self, iters):
        lib().oracle_iterate(self._w, int(iters))

    def scaling(self):
        D, E, c = np.zeros(self.n), np.zeros(self.m), C.c_double()
        lib().oracle_get_scaling(self._w, _pd(D), _pd(E), C.byref(c))
        return D, E, c.value

    def iterate_state(self):
        """(x, z, y) of the current iterate in UNSCALED units, and the current rho."""
        x, z, y, rho = np.zeros(self.n), np.zeros(self.m), np.zeros(self.m), C.c_double()
        lib().oracle_get_iterate(self._w, _pd(x), _pd(z), _pd(y), C.byref(rho))
        D, E, c = self.scaling()
        return D * x, z / E, E * y / c, rho.value

    def solve(self):
        x, y, info = np.zeros(self.n), np.zeros(self.m), Info()
        lib().oracle_solve(self._w, _pd(x), _pd(y), C.byref(info))
        r = _Obj()
        r.x, r.y = x, y
        r.info = _Obj()
        r.info.status_val = info.status
        r.info.status = STATUS[info.status]
        for k in ('iter', 'rho_updates', 'obj_val', 'pri_res', 'dua_res', 'rho_estimate'):
            setattr(r.info, k, getattr(info, k))
        return r
