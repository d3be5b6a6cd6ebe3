#!/usr/bin/env python3
"""One-off fuzz: the two-waves-per-SIMD build of the active-set kernel (1024 < waves <= 2048) against the one-wave build on
random draws: a batch of 6144 must reproduce, bit for bit, the same instances solved as two batches of 3072.
usage: python tools/dev/fuzz_builds.py [n_draws] [first_seed]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, sample_x0
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
WILD = dict(sigma_p=3.0, sigma_v=3.0, max_angle_deg=90.0, sigma_w=3.0)
n_draws = int(sys.argv[1]) if len(sys.argv) > 1 else 10
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
B = 6144
for seed in range(first, first + n_draws):
    rng = np.random.default_rng(52000 + seed)
    N = int(rng.choice([3, 9, 20, 33]))
    mass = float(rng.uniform(0.4, 3.0)); arm = float(rng.uniform(0.1, 0.4)); km = float(rng.uniform(0.005, 0.03)); hov = mass * 9.81 / 4.0
    over = dict(N=N, dt=float(rng.choice([0.02, 0.05, 0.08])), mass=mass, inertia=[float(v) for v in rng.uniform(0.003, 0.03, 3) * mass],
                rotor_x=[arm, 0.0, -arm, 0.0], rotor_y=[0.0, arm, 0.0, -arm], rotor_z=[-km, km, -km, km],
                lbu=[float(hov * rng.uniform(0.0, 0.3))] * 4, ubu=[float(hov * rng.uniform(1.6, 3.5))] * 4,
                W=[float(v) for v in 10.0 ** rng.uniform(-1.5, 1.5, 17)], W_e=[float(v) for v in 10.0 ** rng.uniform(-1, 2, 13)],
                levenberg_marquardt=float(rng.choice([1e-3, 7e-3, 0.1])), sim_num_steps=int(rng.choice([1, 2])),
                flags=_lib.FLAG_TEAM_MAPPING | 1, max_batch=B)
    dist = [NEAR_HOVER, AGGRESSIVE, WILD][int(rng.integers(0, 3))]
    x0 = sample_x0(B, 53000 + seed, **dist)
    yref = np.zeros((N, 17)); yref[:, 2] = 1.0; yref[:, 6] = 1.0; yref[:, 13:] = hov
    ye = yref[0, :13].copy()
    traj = bool(rng.integers(0, 2))
    s = NmpcOcpSolver(_lib.default_config(**over))
    big = s.solve_batch(x0, yref, ye, want_traj=traj)
    same = True
    for h in range(2):
        part = s.solve_batch(x0[h * 3072:(h + 1) * 3072], yref, ye, want_traj=traj)
        for key in ("u0", "status") + (("x", "u") if traj else ()):
            same &= np.array_equal(big[key][h * 3072:(h + 1) * 3072], part[key], equal_nan=True)
    st = s.stats()
    flag = "" if same else "   <-- CHECK"
    bad += bool(flag)
    print(f"seed {seed:3d} N={N:2d} steps={over['sim_num_steps']} dist={'NAW'[[NEAR_HOVER, AGGRESSIVE, WILD].index(dist)]} traj={int(traj)}: equal {same} "
          f"status {np.bincount(big['status'], minlength=5)} passes max {st['polish_max']} ipm max {st['iter_max']}{flag}", flush=True)
    s.close()
print("draws to check:", bad)
