#!/usr/bin/env python3
"""One-off fuzz: the result of an instance must not depend on its wave-mates.  Random draws as tools/dev/fuzz_parity.py; the
batch is solved as drawn and permuted (cold, then warm-started), results compared bit for bit.  GPU only (no oracle).
usage: python tools/dev/fuzz_perm.py [n_draws] [first_seed]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, sample_x0
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
WILD = dict(sigma_p=3.0, sigma_v=3.0, max_angle_deg=90.0, sigma_w=3.0)
n_draws = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for seed in range(first, first + n_draws):
    rng = np.random.default_rng(7000 + seed)
    N = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 20, 24, 31, 40, 57]))
    mass = float(rng.uniform(0.3, 4.0)); arm = float(rng.uniform(0.08, 0.5)); km = float(rng.uniform(0.003, 0.04)); hov = mass * 9.81 / 4.0
    B = int(rng.choice([1, 3, 4, 5, 63, 64, 65, 130, 257, 511]))
    over = dict(N=N, dt=float(rng.choice([0.01, 0.02, 0.05, 0.08, 0.1])), mass=mass,
                inertia=[float(v) for v in rng.uniform(0.002, 0.04, 3) * mass],
                rotor_x=[arm, 0.0, -arm, 0.0], rotor_y=[0.0, arm, 0.0, -arm], rotor_z=[-km, km, -km, km],
                lbu=[float(hov * rng.uniform(0.0, 0.5))] * 4, ubu=[float(hov * rng.uniform(1.3, 4.0))] * 4,
                W=[float(v) for v in 10.0 ** rng.uniform(-2, 2, 17)], W_e=[float(v) for v in 10.0 ** rng.uniform(-1, 2.5, 13)],
                levenberg_marquardt=float(rng.choice([0.0, 1e-4, 7e-3, 0.1, 1.0])), sim_num_steps=int(rng.choice([1, 2, 2, 3])),
                lm_scaled_by_dt=int(rng.integers(0, 2)), cost_scaled_by_dt=int(rng.integers(0, 2)),
                flags=_lib.FLAG_TEAM_MAPPING | int(rng.integers(0, 2)), max_batch=B,
                qp_polish_ckpt=int(rng.choice([0, 1, 4, 12, 100])))
    dist = [NEAR_HOVER, AGGRESSIVE, WILD][int(rng.integers(0, 3))]
    x0 = sample_x0(B, 9000 + seed, **dist)
    _ = bool(rng.integers(0, 2))
    goal = rng.normal(0.0, 1.0, (B, 3)) + np.array([0.0, 0.0, 1.0]); vel = rng.normal(0.0, 0.3, (B, 3))
    yref = np.zeros((B, N, 17)); ye = np.zeros((B, 13))
    for k in range(N + 1):
        row = np.zeros((B, 13)); row[:, 0:3] = goal + vel * (k * over["dt"]); row[:, 3:6] = vel; row[:, 6] = 1.0
        if k < N:
            yref[:, k, :13] = row; yref[:, k, 13:] = hov
        else:
            ye[:] = row
    s = NmpcOcpSolver(_lib.default_config(**over))
    a = s.solve_batch(x0, yref, ye, want_traj=True)
    perm = np.random.default_rng(seed).permutation(B)
    b = s.solve_batch(x0[perm], yref[perm], ye[perm], want_traj=True)
    same = all(np.array_equal(a[k][perm], b[k], equal_nan=True) for k in ("u0", "status", "x", "u"))
    a2 = s.solve_batch(x0, yref, ye, x_init=a["x"], u_init=a["u"], want_traj=True)
    b2 = s.solve_batch(x0[perm], yref[perm], ye[perm], x_init=a["x"][perm], u_init=a["u"][perm], want_traj=True)
    same2 = all(np.array_equal(a2[k][perm], b2[k], equal_nan=True) for k in ("u0", "status", "x", "u"))
    nd = int((np.abs(a["u0"][perm] - b["u0"]).max(1) > 0).sum()) if not same else 0
    if not same and "-v" in sys.argv:
        for key in ("u0", "status", "x", "u"):
            d = np.abs(a[key][perm].astype(float) - b[key].astype(float)).reshape(B, -1).max(1)
            idx = np.nonzero(d > 0)[0]
            print(f"   {key}: {len(idx)} instances differ, max {d.max():.2e}; permuted positions {idx[:8]} (wave {idx[:8] // 4}) original {perm[idx[:8]]}")
        npol = s.stats()
        print("   stats", {k: npol[k] for k in ("polish_max", "iter_max", "n_status")})
    flag = "" if (same and same2) else "   <-- CHECK"
    bad += bool(flag)
    st = s.stats()
    print(f"seed {seed:3d} N={N:2d} B={B:3d} steps={over['sim_num_steps']} share={over['flags'] & 1} dist={'NAW'[[NEAR_HOVER, AGGRESSIVE, WILD].index(dist)]}: cold equal {same} ({nd} instances differ) warm equal {same2}{flag}", flush=True)
    s.close()
print("draws to check:", bad)
