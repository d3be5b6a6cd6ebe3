#!/usr/bin/env python3
"""One-off fuzz of NMPC_DTYPE_F32IO (FP32 buffers, FP64 arithmetic) against the FP64 solver on the same float-representable
inputs, over the random draws of tools/dev/fuzz_parity.py.  usage: python tools/dev/fuzz_f32io.py [n_draws] [first_seed]"""
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
f = lambda a: np.asarray(a, np.float32).astype(np.float64)
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
    x0 = f(sample_x0(B, 9000 + seed, **dist))
    per_inst = bool(rng.integers(0, 2))
    goal = rng.normal(0.0, 1.0, (B, 3)) + np.array([0.0, 0.0, 1.0]); vel = rng.normal(0.0, 0.3, (B, 3))
    yref = np.zeros((B, N, 17)); ye = np.zeros((B, 13))
    for k in range(N + 1):
        row = np.zeros((B, 13)); row[:, 0:3] = goal + vel * (k * over["dt"]); row[:, 3:6] = vel; row[:, 6] = 1.0
        if k < N:
            yref[:, k, :13] = row; yref[:, k, 13:] = hov
        else:
            ye[:] = row
    if not per_inst:
        yref, ye = yref[0], ye[0]
    yref, ye = f(yref), f(ye)
    if over["sim_num_steps"] > 2:
        print(f"seed {seed:3d}: skipped (F32IO is built for <= 2 integrator steps)")
        continue
    s64 = NmpcOcpSolver(_lib.default_config(**over))
    s32 = NmpcOcpSolver(_lib.default_config(**dict(over, dtype=_lib.DTYPE_F32IO)))
    a = s64.solve_batch(x0, yref, ye, want_traj=True)
    b = s32.solve_batch(x0, yref, ye, want_traj=True)
    sm = int((a["status"] != b["status"]).sum())
    ok = (a["status"] == 0) & (b["status"] == 0)
    scale = max(1.0, hov)
    d1 = float(np.abs(a["u0"][ok] - b["u0"][ok]).max()) / scale if ok.any() else 0.0
    xs = max(1.0, float(np.abs(a["x"][ok]).max())) if ok.any() else 1.0
    dx = float(np.abs(a["x"][ok] - b["x"][ok]).max()) / xs if ok.any() else 0.0
    # warm start from the FP32 trajectories on both
    xi, ui = f(b["x"]), f(b["u"])
    a2 = s64.solve_batch(x0, yref, ye, x_init=xi, u_init=ui); b2 = s32.solve_batch(x0, yref, ye, x_init=xi, u_init=ui)
    sm2 = int((a2["status"] != b2["status"]).sum())
    ok2 = (a2["status"] == 0) & (b2["status"] == 0)
    d2 = float(np.abs(a2["u0"][ok2] - b2["u0"][ok2]).max()) / scale if ok2.any() else 0.0
    flag = "" if (sm == 0 and sm2 == 0 and d1 < 1e-6 and d2 < 1e-6 and dx < 1e-6) else "   <-- CHECK"
    bad += bool(flag)
    print(f"seed {seed:3d} N={N:2d} B={B:3d} share={over['flags'] & 1} dist={'NAW'[[NEAR_HOVER, AGGRESSIVE, WILD].index(dist)]} ok {int(ok.sum())}/{B}: "
          f"cold |du0| {d1:.1e} |dx|/|x| {dx:.1e} warm |du0| {d2:.1e} status mismatches {sm}+{sm2}{flag}", flush=True)
    s64.close(); s32.close()
print("draws to check:", bad)
