#!/usr/bin/env python3
"""One-off fuzz of the closed loop (solve -> hold -> plant step -> warm-started solve, rollout.py) on random vehicles and
tunings against the same loop run through the CPU oracle, instance by instance, including the node's behaviour after a
failed solve (re-publish the held command, restart cold).  usage: python tools/dev/fuzz_rollout.py [n] [first_seed]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from oracle import oracle as O
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.rollout import ClosedLoopRollout
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, sample_x0
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
from tests.oracle_solver import OracleOcpSolver
n_draws = int(sys.argv[1]) if len(sys.argv) > 1 else 20
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for seed in range(first, first + n_draws):
    rng = np.random.default_rng(81000 + seed)
    N = int(rng.choice([5, 10, 20, 30]))
    mass = float(rng.uniform(0.4, 2.5)); arm = float(rng.uniform(0.1, 0.35)); km = float(rng.uniform(0.005, 0.03)); hov = mass * 9.81 / 4.0
    over = dict(N=N, dt=float(rng.choice([0.02, 0.05])), mass=mass, inertia=[float(v) for v in rng.uniform(0.004, 0.03, 3) * mass],
                rotor_x=[arm, 0.0, -arm, 0.0], rotor_y=[0.0, arm, 0.0, -arm], rotor_z=[-km, km, -km, km],
                lbu=[float(hov * rng.uniform(0.0, 0.3))] * 4, ubu=[float(hov * rng.uniform(1.5, 3.0))] * 4,
                W=[float(v) for v in 10.0 ** rng.uniform(-1, 1.5, 17)], W_e=[float(v) for v in 10.0 ** rng.uniform(-0.5, 1.5, 13)],
                levenberg_marquardt=float(rng.choice([1e-3, 7e-3, 0.05])), sim_num_steps=int(rng.choice([1, 2])), max_batch=64)
    s = NmpcOcpSolver(_lib.default_config(**over))
    c = OracleOcpSolver(s.config).c; c.qp_polish = 1
    B, steps = 48, 8
    x0 = sample_x0(B, 82000 + seed, **(AGGRESSIVE if rng.integers(0, 2) else NEAR_HOVER))
    sp = tuple(float(v) for v in (rng.normal(0, 0.5, 3) + [0, 0, 1.0])); yaw = float(rng.uniform(-1, 1))
    xs, us = ClosedLoopRollout(s, B).run(x0, steps, setpoint=sp, yaw=yaw)
    yref, ye = O.hover_yref(c, pos=sp, yaw=yaw)
    worst_u = worst_x = 0.0
    for b in rng.choice(B, 3, replace=False):
        x = x0[b].copy(); xt = ut = None; held = np.full(4, hov)
        for t in range(steps):
            r = O.solve_batch(c, x[None], yref, ye, want_traj=True) if xt is None else O.solve_batch(c, x[None], yref, ye, x_init=xt, u_init=ut, want_traj=True)
            xt, ut = r["x"], r["u"]
            if r["status"][0] == 0:
                held = np.clip(r["u0"][0], [c.lbu[i] for i in range(4)], [c.ubu[i] for i in range(4)])
            worst_u = max(worst_u, float(np.abs(us[t, b] - r["u0"][0]).max()))
            x = O.integrate(c, x, held)[0]
            x[6:10] /= np.linalg.norm(x[6:10])
            worst_x = max(worst_x, float(np.abs(xs[t + 1, b] - x).max()))
    flag = "" if (worst_u < 1e-7 and worst_x < 1e-7) else "   <-- CHECK"
    bad += bool(flag)
    print(f"seed {seed:3d} N={N:2d} dt={over['dt']} steps={over['sim_num_steps']}: worst |du0| {worst_u:.1e} worst |dx| {worst_x:.1e} over {steps} ticks{flag}", flush=True)
    s.close()
print("draws to check:", bad)
