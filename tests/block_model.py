"""numpy statement of the parallel-in-time Riccati factorisation of csrc/nmpc_block.hpp (matrix level, homogeneous form xbar = (x, 1)):
test infrastructure - tests/test_block_model.py runs it on the oracle's linearisation of the reference's vehicle, tools/dev/
block_riccati_model.py on random data.

A stage is (Abar, Bm, Qbar, D, rhat): Abar (n x n, last row e') with b_k and the pinned inputs in its last column, Bm the input
matrix with the columns of pinned inputs zeroed, Qbar the state cost with the gradient in its last row / column, D the input
weights, rhat the input gradient (-D v for an input pinned at v)."""
import numpy as np


def stage_step(P, st):
    """one stage of the backward recursion: (P_k, Kbar, H)"""
    A, Bm, Q, D, rhat = st
    X = Bm.T @ P @ A
    X[:, -1] += rhat
    H = np.diag(D) + Bm.T @ P @ Bm
    K = np.linalg.solve(H, X)
    Pn = Q + A.T @ P @ A - X.T @ K
    # kept exactly symmetric, as the kernels keep it: the antisymmetric part of the rounding errors is not contracted by the
    # recursion (one side sees the open loop) and takes the homogeneous form apart within a few hundred stages
    return 0.5 * (Pn + Pn.T), K, H


def sequential(stages, PN):
    """gains of every stage and the value at stage 0"""
    P, Ks = PN.copy(), []
    for st in reversed(stages):
        P, K, _ = stage_step(P, st)
        Ks.append(K)
    return P, Ks[::-1]


def aggregate(stages):
    """(J, Psi, C) of a block: the sweep from a zero terminal value, the product of its closed-loop transitions, their Gramian"""
    n = stages[0][0].shape[0]
    P0, Psi, Cm = np.zeros((n, n)), np.eye(n), np.zeros((n, n))
    for st in reversed(stages):
        A, Bm = st[0], st[1]
        Pn, K, H = stage_step(P0, st)
        G = Psi @ Bm
        Cm += G @ np.linalg.solve(H, G.T)
        Psi = Psi @ (A - Bm @ K)
        P0 = Pn
    return P0, Psi, Cm


def boundary(Pe, agg):
    """P_s = J + Psi' T Psi with T = L (Dp^-1 + L'C L)^-1 L', P_e = L Dp L' (the pivot of the constant replaced by 1)"""
    J, Psi, Cm = agg
    nx = Pe.shape[0] - 1
    Lc = np.linalg.cholesky(Pe[:nx, :nx])
    dp = np.diag(Lc) ** 2
    Lu = Lc / np.diag(Lc)
    L = np.eye(nx + 1)
    L[:nx, :nx] = Lu
    L[nx, :nx] = np.linalg.solve(Lu * dp, Pe[:nx, nx])
    Dp = np.r_[dp, 1.0]
    E = np.diag(1.0 / Dp) + L.T @ Cm @ L
    T = L @ np.linalg.solve(E, L.T)
    return J + Psi.T @ T @ Psi


def block_parallel(stages, PN, J):
    """the three launches: aggregates of blocks 0 .. J-2 (independent), boundary scan (sequential), final sweeps (independent)"""
    N = len(stages)
    M = -(-N // J)
    cuts = list(range(0, N, M)) + [N]
    nb = len(cuts) - 1
    aggs = [aggregate(stages[cuts[j]:cuts[j + 1]]) for j in range(nb - 1)]
    ends = [None] * nb                      # value at the END of block j
    ends[nb - 1] = PN
    Ps_last, Ks_last = sequential(stages[cuts[nb - 1]:], PN)
    starts = [None] * nb
    starts[nb - 1] = Ps_last
    for j in range(nb - 2, -1, -1):
        ends[j] = starts[j + 1]
        starts[j] = boundary(ends[j], aggs[j])
    Ks = [None] * N
    Ks[cuts[nb - 1]:] = Ks_last
    recomputed = [None] * nb
    for j in range(nb - 1):
        recomputed[j], Ks[cuts[j]:cuts[j + 1]] = sequential(stages[cuts[j]:cuts[j + 1]], ends[j])
    return starts, recomputed, Ks
