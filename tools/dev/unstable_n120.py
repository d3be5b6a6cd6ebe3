#!/usr/bin/env python3
"""Diagnostic: the unstable-open-loop configuration of tests/test_gpu_parity.py at a chosen horizon, every GPU path."""
import sys, os
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from oracle import oracle as O
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import sample_x0
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
from tests.oracle_solver import OracleOcpSolver
WILD = dict(sigma_p=3.0, sigma_v=3.0, max_angle_deg=90.0, sigma_w=3.0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
over = dict(N=N, dt=0.1, mass=0.4738978976479069, inertia=[0.0017, 0.006, 0.012],
            rotor_x=[0.4541, 0.0, -0.4541, 0.0], rotor_y=[0.0, 0.4541, 0.0, -0.4541], rotor_z=[-0.0141, 0.0141, -0.0141, 0.0141],
            lbu=[0.0452] * 4, ubu=[2.0395] * 4,
            W=[0.1868, 4.7625, 0.1212, 70.9006, 3.7842, 57.0244, 0.0357, 32.6734, 12.5504, 0.1446, 29.7859, 5.8384, 0.409, 11.6702, 0.8986, 3.7607, 0.0134],
            W_e=[0.127, 13.7797, 13.6895, 2.7246, 3.6848, 4.8743, 0.2726, 140.7136, 0.5401, 23.0029, 17.7842, 0.149, 23.2557],
            levenberg_marquardt=0.0, sim_num_steps=1, lm_scaled_by_dt=1, cost_scaled_by_dt=1, flags=_lib.FLAG_TEAM_MAPPING, max_batch=512)
B = 511
x0 = sample_x0(B, 9021, **WILD)
hov = over["mass"] * 9.81 / 4.0
yref = np.zeros((N, 17)); yref[:, 2] = 1.0; yref[:, 6] = 1.0; yref[:, 13:] = hov
ye = yref[0, :13].copy()
for polish in (1, 0):
    s = NmpcOcpSolver(_lib.default_config(**dict(over, qp_polish=polish)))
    c = OracleOcpSolver(s.config).c; c.qp_polish = polish
    out = s.solve_batch(x0, yref, ye); ref = O.solve_batch(c, x0, yref, ye, nthreads=16)
    st = s.stats()
    bad = np.nonzero((ref["status"] == 0) & (out["status"] != 0))[0]
    print(f"polish={polish}: gpu {np.bincount(out['status'], minlength=5)} oracle {np.bincount(ref['status'], minlength=5)} gpu-only failures {bad[:12]} "
          f"ipm max {st['iter_max']} passes max {st['polish_max']}")
    for i in bad[:3]:
        s1 = NmpcOcpSolver(_lib.default_config(**dict(over, qp_polish=polish, max_batch=4)))
        o1 = s1.solve_batch(x0[i:i + 1], yref, ye); t1 = s1.stats()
        print(f"   inst {i} alone: status {o1['status']} ipm {t1['iter_max']} passes {t1['polish_max']}; oracle iters {ref['iters'][i]}")
