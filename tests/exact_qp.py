"""Independent numpy/scipy solution of the OCP-QP (test helper, not the oracle, not the product).

Condenses the stage-wise QP to the inputs and solves the box-constrained dense problem with
scipy's exact active-set bounded least squares (BVLS) -- a route that shares no code and no
algorithm with the Riccati interior-point method of the oracle or of the HIP kernels.
"""
import numpy as np
from scipy.linalg import cholesky, solve_triangular
from scipy.optimize import lsq_linear

NX, NU = 13, 4


def condense(qp, dx0=None):
    """x_stack = Phi dx0 + G u_stack + c;  returns H, g (in u), G, xfree."""
    A, B, b = qp["A"], qp["B"], qp["b"]
    N = A.shape[0]
    dx0 = np.zeros(NX) if dx0 is None else np.asarray(dx0, float)
    G = np.zeros(((N + 1) * NX, N * NU))
    xf = np.zeros((N + 1, NX))
    xf[0] = dx0
    for k in range(N):
        xf[k + 1] = A[k] @ xf[k] + b[k]
        r0, r1 = (k + 1) * NX, (k + 2) * NX
        G[r0:r1, :] = A[k] @ G[k * NX:(k + 1) * NX, :]
        G[r0:r1, k * NU:(k + 1) * NU] += B[k]
    Q = np.diag(qp["Qd"].ravel())
    R = np.diag(qp["Rd"].ravel())
    H = R + G.T @ Q @ G
    g = qp["r"].ravel() + G.T @ (Q @ xf.ravel() + qp["q"].ravel())
    return H, g, G, xf


def solve_exact(qp, dx0=None, bounded=True):
    """Returns (dx [N+1,13], du [N,4]) of the exact QP solution."""
    H, g, G, xf = condense(qp, dx0)
    N = qp["A"].shape[0]
    if bounded:
        C = cholesky(H, lower=False)                     # H = C'C
        d = -solve_triangular(C, g, trans="T", lower=False)
        res = lsq_linear(C, d, bounds=(qp["lo"].ravel(), qp["hi"].ravel()), method="bvls",
                         tol=1e-15, max_iter=2000)
        u = res.x
        # polish on the identified active set (BVLS returns it exactly; one Newton solve)
        lo, hi = qp["lo"].ravel(), qp["hi"].ravel()
        act_lo, act_hi = res.active_mask < 0, res.active_mask > 0
        free = ~(act_lo | act_hi)
        u = np.where(act_lo, lo, np.where(act_hi, hi, u))
        if free.any():
            rhs = -(g[free] + H[np.ix_(free, ~free)] @ u[~free])
            u[free] = np.linalg.solve(H[np.ix_(free, free)], rhs)
    else:
        u = np.linalg.solve(H, -g)
    x = xf.ravel() + G @ u
    return x.reshape(N + 1, NX), u.reshape(N, NU)
