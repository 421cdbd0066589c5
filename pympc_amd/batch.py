"""Batched controllers: B independent ``MPCController``s of identical (nx, nu, Np, Nc) solved in
one kernel launch per control step (additive surface named in SURVEY.md section 8b).

Semantics per instance are those of the reference class (pyMPC/mpc.py): ``setup()`` builds and
cold-solves, ``update(x, u=None, xref=None)`` refreshes q/l/u and warm-solves, ``output()`` returns
the first optimal input, or ``uref`` for instances whose status is not 'solved' (mpc.py:301-304),
and remembers it as the next u_{-1} (mpc.py:330).
"""
import numpy as np

from .solver import BatchProblem


class BatchMPCController:
    def __init__(self, Ad, Bd, Np=20, Nc=None, x0=None, xref=None, uref=None, uminus1=None,
                 Qx=None, QxN=None, Qu=None, QDu=None,
                 xmin=None, xmax=None, umin=None, umax=None, Dumin=None, Dumax=None,
                 eps_feas=1e6, eps_rel=1e-3, eps_abs=1e-3, device=0, stream=None, SOFT_ON=True, **solver_settings):
        Ad = np.asarray(Ad, dtype=float)
        Bd = np.asarray(Bd, dtype=float)
        if Ad.ndim != 3 or Ad.shape[1] != Ad.shape[2]:
            raise ValueError("Ad should be a stack of square matrices of dimension (B,nx,nx)!")
        B, nx = Ad.shape[0], Ad.shape[1]
        if Bd.ndim != 3 or Bd.shape[0] != B or Bd.shape[1] != nx:
            raise ValueError("Bd should be a stack of matrices of dimension (B,nx,nu)!")
        nu = Bd.shape[2]
        if not Np > 1:
            raise ValueError("Np should be > 1!")
        if Nc is not None and not Nc <= Np:
            raise ValueError("Nc should be <= Np!")
        self.B, self.nx, self.nu, self.Np, self.Nc = B, nx, nu, Np, (Np if Nc is None else Nc)

        def bc(a, shape, default):
            if a is None:
                a = default
            return np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=float), shape))

        inf = np.inf
        self.Ad, self.Bd = Ad, Bd
        self.x0 = bc(x0, (B, nx), 0.0)
        self.uref = bc(uref, (B, nu), 0.0)
        self.uminus1 = bc(uminus1, (B, nu), self.uref)
        if xref is None:
            self.xref = np.zeros((B, nx))
        else:
            xref = np.asarray(xref, dtype=float)
            if xref.shape[-1] != nx:
                raise ValueError("xref should be (B,nx) or (B,Np+1,nx)!")
            self.xref = bc(xref, (B, Np + 1, nx) if (xref.ndim == 3 or (xref.ndim == 2 and xref.shape[0] == Np + 1 and B != Np + 1)) else (B, nx), None)
        self.Qx = bc(Qx, (B, nx, nx), 0.0)
        self.QxN = bc(QxN, (B, nx, nx), self.Qx)
        self.Qu = bc(Qu, (B, nu, nu), 0.0)
        self.QDu = bc(QDu, (B, nu, nu), 0.0)
        self.xmin = bc(xmin, (B, nx), -inf)
        self.xmax = bc(xmax, (B, nx), inf)
        self.umin = bc(umin, (B, nu), -inf)
        self.umax = bc(umax, (B, nu), inf)
        self.Dumin = bc(Dumin, (B, nu), -inf)
        self.Dumax = bc(Dumax, (B, nu), inf)
        self.eps_feas = bc(eps_feas, (B, 1), None)
        self.eps_rel, self.eps_abs = eps_rel, eps_abs
        self.u_failure = self.uref
        self.device, self.stream = device, stream
        self.solver_settings = dict(solver_settings)
        self.SOFT_ON = bool(SOFT_ON)           # pyMPC's hidden switch (mpc.py:237): False = hard state box, no slack variables
        self.prob = None
        self.uminus1_rh = None
        self.x0_rh = None
        self._u_last = None
        self._status = None
        # does the device copy of u_{-1} equal self.uminus1_rh?  output() moves the host value on (mpc.py:330) without
        # touching the device; setup()/update() upload it, step()/run() leave the applied input on the device themselves
        self._um1_on_device = False

    def setup(self, solve=True):
        self.x0_rh = self.x0.copy()
        self.uminus1_rh = self.uminus1.copy()
        # same kwarg swap as the reference (mpc.py:266)
        st = dict(warm_start=True, eps_abs=self.eps_rel, eps_rel=self.eps_abs, soft_constraints=int(self.SOFT_ON))
        st.update(self.solver_settings)
        self.prob = BatchProblem(self.B, self.nx, self.nu, self.Np, self.Nc, device=self.device, stream=self.stream, **st)
        self.prob.setup(self.Ad, self.Bd, self.Qx, self.QxN, self.Qu, self.QDu, self.xmin, self.xmax,
                        self.umin, self.umax, self.Dumin, self.Dumax, self.uref, self.eps_feas,
                        self.x0_rh, self.uminus1_rh, self.xref)
        self._um1_on_device = True
        if solve:
            self.solve()

    def share_factor(self):
        """One model, many states (test_scripts/example_mpc_function.py:105-111): after ``setup()`` of a batch whose instances all carry the same model, every
        instance whose factorization inputs equal instance 0's solves with ONE shared copy of its KKT factor (``mpcqp_share_factor``); scatter the states with
        ``update()`` afterwards.  Results do not change; returns how many instances share (0 on the register-resident backends)."""
        return self.prob.share_factor()

    def update(self, x, u=None, xref=None, solve=True):
        self.x0_rh = x
        if u is not None:
            self.uminus1_rh = u
        if xref is not None:
            self.xref = xref
        self.prob.update(self.x0_rh, self.uminus1_rh, xref)
        self._um1_on_device = True
        if solve:
            self.solve()

    def solve(self):
        self.prob.solve_async()
        self._u_last = None

    def step(self, x, u=None, xref=None):
        """``u = K(x, u_{-1})``: ``update(x, u, xref)`` followed by ``output()`` in one library call
        (MPCController.__controller_function__, mpc.py:377-384)."""
        self.x0_rh = x
        if u is not None:
            self.uminus1_rh = u
        if xref is not None:
            self.xref = xref
        if u is None and not self._um1_on_device:
            u = self.uminus1_rh                   # update(x, u=None) uses the input of the last output() (mpc.py:330,357-359)
        uMPC = self.prob.mpc_step(x, u, xref)
        self.uminus1_rh = uMPC
        self._um1_on_device = True                # mpcqp_mpc_step stored it as the next u_{-1}
        self._u_last = None
        return uMPC

    def run(self, nsteps, w=None, Ap=None, Bp=None, xref_traj=None, estimator=None):
        """``nsteps`` closed-loop steps on the device, equivalent to
        ``for k in range(nsteps): u = K.output(); x = Ap @ x + Bp @ u + w[k]; K.update(x, u, xref_traj[k])``
        (the loop of examples/example_point_mass.py:88-101 with a linear plant; default plant = (Ad, Bd)), or, with
        ``estimator = BatchLinearStateEstimator`` (pympc_amd.kalman), to the output-feedback loop of
        examples/example_inverted_pendulum_kalman.py:135-174 -- the estimator object supplies C, L, the measurement noise
        ``estimator.v`` [nsteps,B,ny] (optional) and the true plant state ``estimator.x_true`` [B,nx], advanced in place.
        Returns ``dict(x=[nsteps+1,B,nx], u=[nsteps,B,nu], status=[nsteps,B] (OSQP status values), iter=[nsteps,B])``
        plus ``xhat`` and ``y`` with an estimator."""
        est = None
        if estimator is not None:
            est = dict(C=estimator.C, L=estimator.L, x_true=estimator.x_true, v=getattr(estimator, 'v', None))
        out = self.prob.mpc_run(nsteps, w=w, Ap=Ap, Bp=Bp, xref_traj=xref_traj, estimator=est)
        xt, ut, st, it = out[:4]
        res = dict(x=xt, u=ut, status=st, iter=it)
        if estimator is not None:
            res['xhat'], res['y'] = out[4], out[5]
            self.x0_rh = out[4][-1].copy()
            estimator.x = out[4][-1].copy()
        else:
            self.x0_rh = xt[-1].copy()
        if xref_traj is not None:
            self.xref = np.asarray(xref_traj)[-1].reshape(self.B, -1)
        self.uminus1_rh = ut[-1].copy()
        self._um1_on_device = True
        self._u_last = None
        return res

    def status(self):
        """Per-instance OSQP status strings of the last solve."""
        infos = self.prob.infos()
        self._infos = infos
        return [self.prob.status_string(i.status) for i in infos]

    def output(self, return_status=False, return_x_seq=False, return_u_seq=False, return_eps_seq=False,
               return_obj_val=False):
        nx, nu, Np, Nc = self.nx, self.nu, self.Np, self.Nc
        want_seq = return_x_seq or return_u_seq or return_eps_seq
        info = {}
        if want_seq:
            x, _, infos = self.prob.solution(want_y=False)
            u0 = x[:, (Np + 1) * nx:(Np + 1) * nx + nu].copy()
        else:
            u0 = self.prob.u0()
            infos = self.prob.infos()
        solved = np.array([i.status == 1 for i in infos])
        uMPC = np.where(solved[:, None], u0, self.u_failure)
        if return_x_seq:
            info['x_seq'] = x[:, :(Np + 1) * nx].reshape(self.B, Np + 1, nx)
        if return_u_seq:
            info['u_seq'] = x[:, (Np + 1) * nx:(Np + 1) * nx + Nc * nu].reshape(self.B, Nc, nu)
        if return_eps_seq:
            o = (Np + 1) * nx + Nc * nu
            info['eps_seq'] = x[:, o:o + (Np + 1) * nx].reshape(self.B, -1, nx)          # (empty without slack variables)
        if return_status:
            info['status'] = [self.prob.status_string(i.status) for i in infos]
            info['iter'] = np.array([i.iter for i in infos])
        if return_obj_val:
            info['obj_val'] = np.array([i.obj_val for i in infos])
        self.uminus1_rh = uMPC
        self._um1_on_device = False
        return uMPC if len(info) == 0 else (uMPC, info)
