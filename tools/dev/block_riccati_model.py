#!/usr/bin/env python3
"""Matrix-level model of the block-parallel Riccati factorisation (csrc/nmpc_block.hpp) - the identities the kernels rely on, checked in
numpy on random stage data of the padded homogeneous shape (xbar = (x, 1): Abar's last row is e', Bbar's is zero, the stage gradient sits
in the last row / column of Qbar, the input gradient in the (u, 1) cross term).

  sequential:  P_k = Qbar + Abar' P Abar - X' H^-1 X,   X = Bm' P Abar + rhat e',   H = D + Bm' P Bm
  block [s, e) with P_e = 0:  J = P^0_s,  Psi = prod of the closed-loop transitions of that sweep,  C = sum (Psi_{k+1} Bm) H^-1 (Psi_{k+1} Bm)'
  any terminal value:   P_s = J + Psi' T Psi,   T = P_e (I + C P_e)^-1 = L (D_p^-1 + L' C L)^-1 L'   with  P_e = L D_p L'
(the last pivot of P_e - the constant of the value function - is replaced by 1: it reaches nothing but the constant of P_s)."""
import numpy as np

rng = np.random.default_rng(0)
nx, nu, M = 13, 4, 24
import sys
SPREAD = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
n = nx + 1


def stage():
    A = np.eye(n); A[:nx, :nx] += SPREAD * rng.normal(size=(nx, nx)); A[:nx, nx] = rng.normal(size=nx)
    B = np.zeros((n, nu)); B[:nx] = rng.normal(size=(nx, nu))
    Q = np.zeros((n, n)); Q[:nx, :nx] = np.diag(rng.uniform(0.1, 10, nx)); q = rng.normal(size=nx); Q[:nx, nx] = q; Q[nx, :nx] = q
    D = rng.uniform(0.5, 2, nu); rhat = rng.normal(size=nu)
    mask = (rng.uniform(size=nu) > 0.3).astype(float)
    vp = rng.normal(size=nu) * (1 - mask)
    A = A.copy(); A[:, nx] += B @ vp                 # pinned inputs enter through b
    rhat = np.where(mask > 0, rhat, -D * vp)
    return A, B * mask, Q, D, rhat


def step(P, st):
    A, Bm, Q, D, rhat = st
    X = Bm.T @ P @ A; X[:, nx] += rhat
    H = np.diag(D) + Bm.T @ P @ Bm
    K = np.linalg.solve(H, X)
    return Q + A.T @ P @ A - X.T @ K, K, H


stages = [stage() for _ in range(M)]
Pe = np.zeros((n, n)); W = rng.normal(size=(nx, nx)); Pe[:nx, :nx] = W @ W.T + np.eye(nx); p = rng.normal(size=nx); Pe[:nx, nx] = p; Pe[nx, :nx] = p; Pe[nx, nx] = 3.0
# sequential
P = Pe.copy()
for st in reversed(stages):
    P, _, _ = step(P, st)
# block aggregate
P0 = np.zeros((n, n)); Psi = np.eye(n); C = np.zeros((n, n))
for st in reversed(stages):
    A, Bm, Q, D, rhat = st
    Pn, K, H = step(P0, st)
    G = Psi @ Bm
    C += G @ np.linalg.solve(H, G.T)
    Psi = Psi @ (A - Bm @ K)
    P0 = Pn
# boundary formula through two LDL' factorisations
Ph = Pe.copy()
Lp = np.linalg.cholesky(Ph[:nx, :nx]); dp = np.diag(Lp) ** 2; Lp = Lp / np.diag(Lp)
L = np.eye(n); L[:nx, :nx] = Lp; L[nx, :nx] = np.linalg.solve(Lp * dp, p)       # last row of the unit factor; last pivot := 1
Dp = np.r_[dp, 1.0]
E = np.diag(1 / Dp) + L.T @ C @ L
T = L @ np.linalg.solve(E, L.T)
Ps = P0 + Psi.T @ T @ Psi
err = np.abs(Ps - P); err[nx, nx] = 0
print("max |P_s(block) - P_s(sequential)| off the constant:", err.max(), " scale", np.abs(P).max())
assert err.max() < 1e-9 * np.abs(P).max()
print("hom row/col of C:", np.abs(C[nx]).max(), np.abs(C[:, nx]).max(), " last row of Psi:", Psi[nx])
