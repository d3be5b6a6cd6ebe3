#!/usr/bin/env python3
"""One-off fuzz of the GPU path against the CPU oracle over random vehicles, tunings, horizons, batch sizes, references and
warm starts (the unit test tests/test_gpu_parity.py::test_randomised_vehicle_tuning_and_references holds 8 such draws).
usage: python tools/dev/fuzz_parity.py [n_draws] [first_seed]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from oracle import oracle as O  # noqa: E402
from rotors_mpc_controller_amd import _lib  # noqa: E402
from rotors_mpc_controller_amd.solver import NmpcOcpSolver  # noqa: E402
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, sample_x0  # noqa: E402
from tests.oracle_solver import OracleOcpSolver  # noqa: E402

WILD = dict(sigma_p=3.0, sigma_v=3.0, max_angle_deg=90.0, sigma_w=3.0)
n_draws = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
MAPPING = "lane" if "--lane" in sys.argv else ("cond" if "--cond" in sys.argv else "team")     # the fidelity kernels: plain IPM on both sides
worst = 0.0
bad = 0
for seed in range(first, first + n_draws):
    rng = np.random.default_rng(7000 + seed)
    N = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 20, 24, 31, 40, 57]))
    mass = float(rng.uniform(0.3, 4.0))
    arm = float(rng.uniform(0.08, 0.5))
    km = float(rng.uniform(0.003, 0.04))
    hov = mass * 9.81 / 4.0
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
    if MAPPING != "team":
        over.pop("qp_polish_ckpt")
        over.update(qp_polish=0, flags=(over["flags"] & 1) | (_lib.FLAG_CONDENSED_QP if MAPPING == "cond" else 0))
        if MAPPING == "cond":
            over.update(qp_cond_N=int(rng.choice([2, 3, 5])))
            if N > 40 or over["sim_num_steps"] > 2:
                print(f"seed {seed:3d}: skipped (condensed fidelity kernel: N <= 40)")
                continue
        B = min(B, 130); over["max_batch"] = B
    s = NmpcOcpSolver(_lib.default_config(**over))
    c = OracleOcpSolver(s.config).c
    c.qp_polish = 1 if MAPPING == "team" else 0
    if MAPPING == "cond":
        c.qp_cond_N = over["qp_cond_N"]
    dist = [NEAR_HOVER, AGGRESSIVE, WILD][int(rng.integers(0, 3))]
    x0 = sample_x0(B, 9000 + seed, **dist)
    per_inst = bool(rng.integers(0, 2))
    goal = rng.normal(0.0, 1.0, (B, 3)) + np.array([0.0, 0.0, 1.0])
    vel = rng.normal(0.0, 0.3, (B, 3))
    yref = np.zeros((B, N, 17)); ye = np.zeros((B, 13))
    for k in range(N + 1):
        row = np.zeros((B, 13)); row[:, 0:3] = goal + vel * (k * over["dt"]); row[:, 3:6] = vel; row[:, 6] = 1.0
        if k < N:
            yref[:, k, :13] = row; yref[:, k, 13:] = hov
        else:
            ye[:] = row
    if not per_inst:
        yref, ye = yref[0], ye[0]
    traj = bool(rng.integers(0, 2))
    out = s.solve_batch(x0, yref, ye, want_traj=True)
    ref = O.solve_batch(c, x0, yref, ye, want_traj=True, nthreads=16)
    sm = int((out["status"] != ref["status"]).sum())
    ok = (ref["status"] == 0) & (out["status"] == 0)
    scale = max(1.0, hov)
    d1 = float(np.abs(out["u0"][ok] - ref["u0"][ok]).max()) / scale if ok.any() else 0.0
    dx = float(np.abs(out["x"][ok] - ref["x"][ok]).max()) / scale if ok.any() else 0.0
    if not traj:
        o1 = s.solve_batch(x0, yref, ye)
        assert np.array_equal(o1["u0"], out["u0"]) and np.array_equal(o1["status"], out["status"]), "u0 differs with / without trajectories"
    if "--random-init" in sys.argv:      # an arbitrary (dynamically inconsistent) linearisation trajectory, the same on both sides
        xi = np.tile(x0[:, None, :], (1, N + 1, 1)) + rng.normal(0, 0.3, (B, N + 1, 13)); xi[:, 0] = x0
        qn = np.linalg.norm(xi[:, :, 6:10], axis=2, keepdims=True); xi[:, :, 6:10] /= np.where(qn > 0, qn, 1.0)
        ui = rng.uniform(over["lbu"][0], over["ubu"][0], (B, N, 4))
        out2 = s.solve_batch(x0, yref, ye, x_init=xi, u_init=ui, want_traj=True)
        ref2 = O.solve_batch(c, x0, yref, ye, x_init=xi, u_init=ui, want_traj=True, nthreads=16)
    else:
        out2 = s.solve_batch(x0, yref, ye, x_init=out["x"], u_init=out["u"], want_traj=True)
        ref2 = O.solve_batch(c, x0, yref, ye, x_init=ref["x"], u_init=ref["u"], want_traj=True, nthreads=16)
    sm2 = int((out2["status"] != ref2["status"]).sum())
    ok2 = ok & (ref2["status"] == 0) & (out2["status"] == 0)
    d2 = float(np.abs(out2["u0"][ok2] - ref2["u0"][ok2]).max()) / scale if ok2.any() else 0.0
    st = s.stats()
    flag = "" if (sm == 0 and sm2 == 0 and d1 < 1e-9 and d2 < 1e-8 and dx < 1e-8) else "   <-- CHECK"
    bad += bool(flag)
    worst = max(worst, d1, d2)
    print(f"seed {seed:3d} N={N:2d} B={B:3d} steps={over['sim_num_steps']} share={over['flags'] & 1} ckpt={over.get('qp_polish_ckpt', 0):3d} "
          f"dist={'NAW'[[NEAR_HOVER, AGGRESSIVE, WILD].index(dist)]} ok {int(ok.sum())}/{B}: cold |du0| {d1:.1e} |dx| {dx:.1e} "
          f"warm |du0| {d2:.1e} status mismatches {sm}+{sm2} passes max {st['polish_max']} ipm max {st['iter_max']}{flag}", flush=True)
    s.close()
print(f"worst relative |du0| {worst:.2e}; draws to check: {bad}")
