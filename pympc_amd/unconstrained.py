"""The MPC law without inequality constraints, as gain matrices -- the quantities of the reference's side script
test_scripts/alternative/unconstrained.py:170-183 (``k_x0, k_Xref, k_Uref, k_uminus1``) and doc/latex/main.tex:535-705, computed ON THE
DEVICE by the block-tridiagonal KKT backend.

Without inequality constraints the optimal input sequence is linear in (x0, xref, uref, u_{-1}):

    U* = K_x0 x0 + K_xref xref + K_uref uref + K_um1 u_{-1}          (constant reference xref; U* = (u_0 .. u_{Nc-1}) stacked)

so the columns of the gains are the solutions of the EQUALITY-constrained QP (dynamics rows only) for the unit vectors of those arguments:
one batch of 2 (nx + nu) right-hand sides.  The reference gets them from dense condensing (prediction matrices, a Np nu x Np nu
normal-equation solve).  Here (``mpcqp_eq_solve``): ONE factorization of ``K = P + sigma I + rho_eq A_eq' A_eq`` per column instance -- the
reduced KKT matrix the ADMM kernels use, with weight on the equality rows only: every other row has infinite bounds and gets OSQP's
rho_min -- and then the method of multipliers in residual form,

    r = -(P w + q + A_eq' y) - rho_eq A_eq' (A_eq w - b),    w += K^-1 r,    y += rho_eq (A_eq w - b),

each sweep one batched KKT solve that contracts the error by about ``|P| / rho_eq`` (measured at rho_eq = 1e5: a factor 1e3 per sweep
on the random models, 10 on the unstable cart-pole): every column sweeps until its corrections vanish (1e-13 relative), 5 to 14
solves, all inside ONE kernel launch; against the condensed closed form the gains are 1e-13 or better -- not an ADMM run with OSQP's default rho
schedule (round 3: up to 400 000 iterations to 1e-7).  With the residual formed by the exact operators the fixed point is the KKT
point itself, however much rounding error the stored factor carries at that rho; it only preconditions.  (The KKT residuals the call
returns bottom out near eps_machine * rho_eq * |x| -- the multiplier carries the rounding noise of ``A_eq w - b`` times rho_eq -- while
the inputs are two to four digits better: they are a sanity bound here, the stopping rule is the answer itself.)

This is a utility next to the controller, not a shortcut inside it: ``MPCController.output()`` always reports what the ADMM solve of the
constrained QP returns (status, iteration count, iterate), like the reference's OSQP-backed class does."""
import numpy as np

from .solver import BatchProblem

RHO = 100.0          # rho_eq = 1e3 rho = 1e5 (OSQP's RHO_EQ_OVER_RHO_INEQ)
TOL = 1e-13          # a column stops once a sweep's correction is below this (relative to the largest entry of its solution)
MAX_SWEEPS = 40
RES_SANITY = 1e-6    # KKT residuals (relative) the converged answer must satisfy


class GainSolver:
    """Device handle for the gains of controllers of one shape (nx, nu, Np, Nc): kept by ``MPCController.unconstrained_gains`` so that a
    second call costs one setup kernel and a handful of solves, no allocation."""

    def __init__(self, nx, nu, Np, Nc, tol=TOL, device=0):
        self.nx, self.nu, self.Np, self.Nc, self.tol = nx, nu, Np, Nc, tol
        self.B = 2 * (nx + nu)
        self.prob = BatchProblem(self.B, nx, nu, Np, Nc, device=device, rho=RHO, adaptive_rho=0, alpha=1.0)      # (alpha: only the CPU twin's sweeps, which are ADMM iterations, read it)
        B = self.B
        self.x0, self.xref, self.uref, self.um1 = np.zeros((B, nx)), np.zeros((B, nx)), np.zeros((B, nu)), np.zeros((B, nu))
        self.x0[np.arange(nx), np.arange(nx)] = 1.0
        self.xref[nx + np.arange(nx), np.arange(nx)] = 1.0
        self.uref[2 * nx + np.arange(nu), np.arange(nu)] = 1.0
        self.um1[2 * nx + nu + np.arange(nu), np.arange(nu)] = 1.0

    def gains(self, Ad, Bd, Qx, QxN, Qu, QDu):
        nx, nu, B = self.nx, self.nu, self.B
        bc = lambda M, shape: np.broadcast_to(np.asarray(M, dtype=float), (B,) + shape)
        inf = np.inf
        p = self.prob
        p.setup(bc(Ad, (nx, nx)), bc(Bd, (nx, nu)), bc(Qx, (nx, nx)), bc(QxN, (nx, nx)), bc(Qu, (nu, nu)), bc(QDu, (nu, nu)),
                np.full((B, nx), -inf), np.full((B, nx), inf), np.full((B, nu), -inf), np.full((B, nu), inf), np.full((B, nu), -inf), np.full((B, nu), inf),
                self.uref, np.full((B, 1), 1e6), self.x0, self.um1, self.xref)
        ou = (self.Np + 1) * nx
        res = p.eq_solve(MAX_SWEEPS, cold=True, tol=self.tol)        # ONE launch: every column sweeps until its corrections vanish
        self.sweeps = int(res[:, 4].max())
        self.residuals = np.maximum(res[:, 0] / np.maximum(1.0, res[:, 1]), res[:, 2] / np.maximum(1.0, res[:, 3]))
        x, _, _ = p.solution(want_y=False)
        U = x[:, ou:ou + self.Nc * nu].T.copy()                      # column j = U* for the j-th unit argument
        if not np.isfinite(U).all() or not (self.residuals.max() <= RES_SANITY):
            raise RuntimeError('unconstrained_gains: the multiplier sweeps did not converge (KKT residuals %.2e after %d sweeps)' % (self.residuals.max(), self.sweeps))
        if any(i.status != 1 for i in p.infos()):                    # ('maximum iterations reached': a column did not settle within MAX_SWEEPS)
            raise RuntimeError('unconstrained_gains: the multiplier sweeps did not settle within %d sweeps' % self.sweeps)
        return dict(K_x0=U[:, :nx].copy(), K_xref=U[:, nx:2 * nx].copy(), K_uref=U[:, 2 * nx:2 * nx + nu].copy(), K_um1=U[:, 2 * nx + nu:].copy())


def unconstrained_gains(Ad, Bd, Np, Nc=None, Qx=None, QxN=None, Qu=None, QDu=None, tol=TOL, device=0, solver=None):
    """Returns ``dict(K_x0 [Nc*nu, nx], K_xref [Nc*nu, nx], K_uref [Nc*nu, nu], K_um1 [Nc*nu, nu])``; the first nu rows are the
    feedback law of the receding-horizon controller, u_0 = K_x0[:nu] x0 + ...  Raises ``RuntimeError`` if the sweeps do not converge.
    ``solver``: a ``GainSolver`` of the same shape to reuse (else one is made for the call)."""
    Ad, Bd = np.asarray(Ad, dtype=float), np.asarray(Bd, dtype=float)
    nx, nu = Bd.shape
    Nc = Np if Nc is None else Nc
    z = lambda M, k: np.zeros((k, k)) if M is None else np.asarray(M, dtype=float)
    Qx = z(Qx, nx)
    if solver is None:
        solver = GainSolver(nx, nu, Np, Nc, tol=tol, device=device)
    return solver.gains(Ad, Bd, Qx, Qx if QxN is None else np.asarray(QxN, dtype=float), z(Qu, nu), z(QDu, nu))
