#!/usr/bin/env python3
"""Device time and accuracy of the parallel-in-time Riccati factorisation (csrc/nmpc_block.hpp) against the sequential sweep in the same
code (blocks = 1), on the LQ problems a long-horizon solve ends on.  usage: python tools/block_factor_rate.py [N] [B] [J ...]"""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("NMPC_TEAM_LSTG", "0")
import torch  # noqa: E402

from rotors_mpc_controller_amd import _lib  # noqa: E402
from rotors_mpc_controller_amd.solver import NmpcOcpSolver  # noqa: E402
from rotors_mpc_controller_amd.synthetic import NEAR_HOVER, hover_reference, sample_x0  # noqa: E402
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100
Js = [int(a) for a in sys.argv[3:]] or [1, 5, 10, 15, 20, 25, 30, 40, 60]
s = NmpcOcpSolver(_lib.default_config(N=N, max_batch=B, flags=_lib.FLAG_TEAM_MAPPING))
yref, ye = hover_reference(N, s.config.mass * s.config.gravity / 4.0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
d_x0, d_yr, d_ye = dev(sample_x0(B, 5, **NEAR_HOVER)), dev(yref), dev(ye)
d_u0 = torch.zeros(B, 4, dtype=torch.float64, device="cuda")
d_st = torch.zeros(B, dtype=torch.int32, device="cuda")
s.solve_batch_device(B, d_x0.data_ptr(), d_yr.data_ptr(), d_ye.data_ptr(), True, d_u0.data_ptr(), d_st.data_ptr())
torch.cuda.synchronize()
st, ps = d_st.cpu().numpy(), s.passes(B)
acc = (st == 0) & (ps > 0)
print(f"N = {N}, B = {B}: {int(acc.sum())} instances ended on an accepted pass (passes mean {ps[acc].mean():.2f} max {ps[acc].max()}); "
      f"{int((B + 3) // 4)} waves per block")
fac = torch.zeros(B, N, 80, dtype=torch.float64, device="cuda")
ref = None
for J in Js:
    best = None
    for rep in range(5):
        Jeff, ms = s.block_factor_device(B, J, d_x0.data_ptr(), d_yr.data_ptr(), d_ye.data_ptr(), True, factors_ptr=fac.data_ptr(), timed=True)
        tot = sum(ms)
        if best is None or tot < best[0]:
            best = (tot, ms)
    torch.cuda.synchronize()
    f = fac.cpu().numpy()
    if ref is None:
        ref = f.copy()
    scale = np.abs(ref[acc]).max(axis=(1, 2), keepdims=True)
    err = (np.abs(f[acc] - ref[acc]) / scale).max()
    ms = best[1]
    print(f"blocks {Jeff:3d} ({(N + Jeff - 1) // Jeff:3d} stages each): launch 1 {ms[0]:7.3f} ms | scan {ms[1]:7.3f} ms | launch 3 {ms[2]:7.3f} ms | "
          f"total {best[0]:7.3f} ms | max rel |factor - sequential| {err:.1e}", flush=True)
s.close()
