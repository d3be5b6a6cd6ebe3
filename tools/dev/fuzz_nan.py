#!/usr/bin/env python3
"""One-off fuzz of failure isolation: random draws as tools/dev/fuzz_parity.py with NaN / Inf planted in x0, yref or the warm
start of a few instances.  Those instances must come back with a non-zero status, a zero command and the cold-start point;
every other instance must be bit-identical to the run without the poison.  usage: python tools/dev/fuzz_nan.py [n] [first]"""
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
    rng = np.random.default_rng(61000 + seed)
    N = int(rng.choice([1, 3, 9, 20, 31]))
    mass = float(rng.uniform(0.4, 3.0)); arm = float(rng.uniform(0.1, 0.4)); km = float(rng.uniform(0.005, 0.03)); hov = mass * 9.81 / 4.0
    B = int(rng.choice([5, 64, 130, 257, 1030]))
    over = dict(N=N, dt=float(rng.choice([0.02, 0.05, 0.08])), mass=mass, inertia=[float(v) for v in rng.uniform(0.003, 0.03, 3) * mass],
                rotor_x=[arm, 0.0, -arm, 0.0], rotor_y=[0.0, arm, 0.0, -arm], rotor_z=[-km, km, -km, km],
                lbu=[float(hov * rng.uniform(0.0, 0.3))] * 4, ubu=[float(hov * rng.uniform(1.6, 3.5))] * 4,
                W=[float(v) for v in 10.0 ** rng.uniform(-1.5, 1.5, 17)], W_e=[float(v) for v in 10.0 ** rng.uniform(-1, 2, 13)],
                levenberg_marquardt=float(rng.choice([1e-3, 7e-3, 0.1])), sim_num_steps=int(rng.choice([1, 2, 3])),
                flags=_lib.FLAG_TEAM_MAPPING | int(rng.integers(0, 2)), max_batch=B, qp_polish=int(rng.choice([1, 1, 0])))
    dist = [NEAR_HOVER, AGGRESSIVE, WILD][int(rng.integers(0, 3))]
    x0 = sample_x0(B, 62000 + seed, **dist)
    yref = np.zeros((B, N, 17)); yref[:, :, 2] = 1.0; yref[:, :, 6] = 1.0; yref[:, :, 13:] = hov
    ye = yref[:, 0, :13].copy()
    s = NmpcOcpSolver(_lib.default_config(**over))
    clean = s.solve_batch(x0, yref, ye, want_traj=True)
    warm = bool(rng.integers(0, 2))
    xi, ui = clean["x"].copy(), clean["u"].copy()
    ref = s.solve_batch(x0, yref, ye, x_init=xi, u_init=ui, want_traj=True) if warm else clean
    victims = rng.choice(B, size=min(B, int(rng.integers(1, 5))), replace=False)
    xp, yp, xip = x0.copy(), yref.copy(), xi.copy()
    kinds = []
    for v in victims:
        kind = int(rng.integers(0, 3 if warm else 2))
        val = np.nan if "--nan-only" in sys.argv else [np.nan, np.inf, -np.inf][int(rng.integers(0, 3))]
        if kind == 0: xp[v, int(rng.integers(0, 13))] = val
        elif kind == 1:
            # a reference entry the solution depends on: an input reference of any stage, or a state reference of a stage
            # k >= 1 (the stage-0 state is pinned to x0: its reference only reaches the costate of stage 0)
            kk = int(rng.integers(0, N))
            yp[v, kk, int(rng.integers(13, 17)) if kk == 0 else int(rng.integers(0, 17))] = val
        else: xip[v, int(rng.integers(1, N + 1)), int(rng.integers(0, 13))] = val
        kinds.append(kind)
    out = s.solve_batch(xp, yp, ye, x_init=xip, u_init=ui, want_traj=True) if warm else s.solve_batch(xp, yp, ye, want_traj=True)
    others = np.setdiff1d(np.arange(B), victims)
    iso = all(np.array_equal(out[k][others], ref[k][others], equal_nan=True) for k in ("u0", "status", "x", "u"))
    vic_ok = bool((out["status"][victims] != 0).all() and (out["u0"][victims] == 0).all() and (out["u"][victims] == 0).all())
    # the hand-back of a failed instance: x_k = x0 (whatever x0 is, NaN included), u_k = 0
    hand = all(np.array_equal(out["x"][v], np.tile(xp[v], (N + 1, 1)), equal_nan=True) for v in victims)
    fin_others = bool(np.isfinite(out["u0"][others]).all())
    flag = "" if (iso and vic_ok and hand and fin_others) else "   <-- CHECK"
    bad += bool(flag)
    print(f"seed {seed:3d} N={N:2d} B={B:4d} polish={over['qp_polish']} share={over['flags'] & 1} warm={int(warm)} victims {victims.tolist()} kinds {kinds}: "
          f"others identical {iso}, victims status {out['status'][victims].tolist()} zero command {vic_ok}, cold-start hand-back {hand}{flag}", flush=True)
    s.close()
print("draws to check:", bad)
