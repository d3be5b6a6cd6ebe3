#!/usr/bin/env python3
"""The identities behind csrc/nmpc_block.hpp on RANDOM stage data of the padded homogeneous shape (tests/block_model.py holds the
numpy statement; tests/test_block_model.py runs it on the oracle's linearisation of the reference's vehicle).  The spread of the
random open loop is the argument: the block form loses digits with the instability of the plant (rho ~ 1 + 8 spread).
usage: python tools/dev/block_riccati_model.py [spread]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from tests import block_model as bm  # noqa: E402

rng = np.random.default_rng(0)
nx, nu, N, J = 13, 4, 96, 4
n = nx + 1
SPREAD = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1


def stage():
    A = np.eye(n); A[:nx, :nx] += SPREAD * rng.normal(size=(nx, nx)); A[:nx, nx] = rng.normal(size=nx)
    B = np.zeros((n, nu)); B[:nx] = rng.normal(size=(nx, nu))
    Q = np.zeros((n, n)); Q[:nx, :nx] = np.diag(rng.uniform(0.1, 10, nx)); q = rng.normal(size=nx); Q[:nx, nx] = q; Q[nx, :nx] = q
    D = rng.uniform(0.5, 2, nu); rhat = rng.normal(size=nu)
    mask = (rng.uniform(size=nu) > 0.3).astype(float)
    vp = rng.normal(size=nu) * (1 - mask)
    A[:, nx] += B @ vp                               # pinned inputs enter through b
    return A, B * mask, Q, D, np.where(mask > 0, rhat, -D * vp)


stages = [stage() for _ in range(N)]
PN = np.zeros((n, n)); W = rng.normal(size=(nx, nx)); PN[:nx, :nx] = W @ W.T + np.eye(nx); p = rng.normal(size=nx); PN[:nx, nx] = p; PN[nx, :nx] = p
P0, Ks = bm.sequential(stages, PN)
starts, _, Kb = bm.block_parallel(stages, PN, J)
d = np.abs(starts[0] - P0); d[nx, nx] = 0
print(f"spread {SPREAD}: max |P_0(block) - P_0(sequential)| off the constant {d.max():.2e} (scale {np.abs(P0).max():.2e}); "
      f"max gain difference {max(np.abs(a - b).max() for a, b in zip(Ks, Kb)):.2e}")
