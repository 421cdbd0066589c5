"""Call-compatible stand-in for the reference's older controller ``pyMPC/mpc_no_slack.py`` (hard state constraints, no
control horizon, ``step()`` instead of ``output()``).

It is a thin adapter over ``pympc_amd.MPCController`` with the hidden switch ``SOFT_ON = False`` (mpc.py:237): for one
input that is the very QP ``mpc_no_slack.py:225-291`` builds -- same variables ``[x_0..x_Np | u_0..u_{Np-1}]``, same row
blocks (dynamics | state and input box | Delta-u) -- solved on the GPU at the tolerances the reference hard-codes
(``eps_abs = eps_rel = 1e-4``, mpc_no_slack.py:119).  Behaviour kept: ``setup()`` does not solve, ``step()`` solves and
RAISES unless the status is 'solved' (mpc_no_slack.py:126), ``update(x, u=None)`` refreshes the problem without solving,
``__controller_function__`` does not touch ``uminus1_rh``.  For more than one input the reference's own Delta-u block is
dimensionally inconsistent (one row for the first step, mpc_no_slack.py:270: the solver setup fails there); this adapter
says so up front.
"""
import numpy as np

from .controller import MPCController as _Controller


class MPCController:
    def __init__(self, Ad, Bd, Np=10, x0=None, xref=None, uref=None, uminus1=None, Qx=None, QxN=None, Qu=None, QDu=None,
                 xmin=None, xmax=None, umin=None, umax=None, Dumin=None, Dumax=None):
        self._K = _Controller(Ad, Bd, Np=Np, x0=x0, xref=xref, uref=uref, uminus1=uminus1, Qx=Qx, QxN=QxN, Qu=Qu, QDu=QDu,
                              xmin=xmin, xmax=xmax, umin=umin, umax=umax, Dumin=Dumin, Dumax=Dumax, eps_rel=1e-4, eps_abs=1e-4)
        self._K.SOFT_ON = False

    # the attributes scripts read or set on the reference object live on the wrapped controller
    def __getattr__(self, name):
        return getattr(self.__dict__['_K'], name)

    def __setattr__(self, name, value):
        if name == '_K':
            self.__dict__[name] = value
        else:
            setattr(self._K, name, value)

    def setup(self):
        if self._K.nu != 1:
            raise ValueError("mpc_no_slack builds its Delta-u rows for a single input (mpc_no_slack.py:270); use pympc_amd.MPCController")
        self._K.setup(solve=False)

    def _solve(self):
        self._K.res = self._K.prob.solve()
        if self._K.res.info.status != 'solved':
            raise ValueError('OSQP did not solve the problem!')
        Np, nx, nu = self._K.Np, self._K.nx, self._K.nu
        return self._K.res.x[(Np + 1) * nx:(Np + 1) * nx + nu]          # = res.x[-Np*nu : -(Np-1)*nu]

    def step(self):
        uMPC = self._solve()
        self._K.uminus1_rh = uMPC
        return uMPC

    def update(self, x, u=None):
        self._K.update(x, u, solve=False)

    def __controller_function__(self, x, u):
        self._K.update(x, u, solve=False)
        return self._solve()
