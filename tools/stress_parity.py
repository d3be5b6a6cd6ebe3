#!/usr/bin/env python3
"""Large-sample parity sweep of the GPU path against the CPU oracle (run by hand on the GPU box:
`python tools/stress_parity.py`).  Not collected by pytest: the unit tests hold the same bar on
smaller samples.  Cold start and one warm-started second iteration, three input distributions,
several seeds, shared and per-stage linearisation."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from oracle import oracle as O  # noqa: E402
from rotors_mpc_controller_amd import _lib  # noqa: E402
from rotors_mpc_controller_amd.solver import NmpcOcpSolver  # noqa: E402
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, hover_reference, sample_x0  # noqa: E402
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)

WILD = dict(sigma_p=3.0, sigma_v=3.0, max_angle_deg=90.0, sigma_w=3.0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
POLISH = 0 if "--no-polish" in sys.argv else 1        # --no-polish: the plain interior-point path on both sides
worst = 0.0
bad = 0
for share in (1, 0):
    s = NmpcOcpSolver(_lib.default_config(max_batch=B, flags=share | _lib.FLAG_TEAM_MAPPING, qp_polish=POLISH))
    c = O.default_config(qp_gamma=0.0, qp_polish=POLISH)
    yref, ye = hover_reference(s.config.N, s.config.mass * s.config.gravity / 4.0)
    for name, dist in (("near", NEAR_HOVER), ("aggr", AGGRESSIVE), ("wild", WILD)):
        for seed in (11, 12, 13):
            x0 = sample_x0(B, seed, **dist)
            t = time.time()
            g1 = s.solve_batch(x0, yref, ye, want_traj=True)
            st = s.stats()
            r1 = O.solve_batch(c, x0, yref, ye, want_traj=True)
            ok = (g1["status"] == 0) & (r1["status"] == 0)
            d1 = np.abs(g1["u0"][ok] - r1["u0"][ok]).max()
            dx = np.abs(g1["x"][ok] - r1["x"][ok]).max()
            x1 = x0 + np.random.default_rng(seed).normal(0, 0.01, x0.shape)
            g2 = s.solve_batch(x1, yref, ye, x_init=g1["x"], u_init=g1["u"], want_traj=True)
            r2 = O.solve_batch(c, x1, yref, ye, x_init=g1["x"], u_init=g1["u"], want_traj=True)
            ok2 = (g2["status"] == 0) & (r2["status"] == 0)
            d2 = np.abs(g2["u0"][ok2] - r2["u0"][ok2]).max()
            nst = int((g1["status"] != r1["status"]).sum() + (g2["status"] != r2["status"]).sum())
            worst = max(worst, d1, d2)
            bad += nst
            print(f"share={share} {name} seed {seed}: cold max|du0| {d1:.2e} max|dx| {dx:.2e}  warm max|du0| {d2:.2e}  "
                  f"status mismatches {nst}  gpu status hist {np.bincount(g1['status'], minlength=5).tolist()} "
                  f"passes mean {st['polish_mean']:.3f} max {st['polish_max']} ipm mean {st['iter_mean']:.3f}  ({time.time() - t:.1f} s)", flush=True)
print(f"worst |du0| {worst:.3e}, status mismatches {bad}")
sys.exit(0 if (worst < 1e-8 and bad == 0) else 1)
