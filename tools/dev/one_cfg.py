#!/usr/bin/env python3
"""Dev: one configuration of tools/dev/qp_kernel_check.py in its own process. usage: one_cfg.py steps share polish traj B dist_seed"""
import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from oracle import oracle as O
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, hover_reference, sample_x0
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
steps, share, polish, traj, B, seed = [int(a) for a in sys.argv[1:7]]
s = NmpcOcpSolver(_lib.default_config(N=20, max_batch=B, sim_num_steps=steps, qp_polish=polish, flags=_lib.FLAG_TEAM_MAPPING | share))
yref, ye = hover_reference(20, 0.68 * 9.81 / 4)
x0 = sample_x0(B, seed, **AGGRESSIVE)
o = s.solve_batch(x0, yref, ye, want_traj=bool(traj))
r = O.solve_batch(O.default_config(qp_gamma=0.0, qp_polish=polish, sim_num_steps=steps), x0, yref, ye)
print("ok", sys.argv[1:], np.bincount(o["status"], minlength=5), "|du0|", np.abs(o["u0"] - r["u0"]).max(), "iters equal", np.array_equal(s.iterations(), r["iters"]), flush=True)
