#!/usr/bin/env python3
"""One-off fuzz of the PositionNMPC facade on the GPU against the same facade on the CPU oracle: random vehicle / solver
parameters (what dynamic_reconfigure can set, controller.py:63-172), random states, a moving setpoint, several ticks with the
unshifted warm start (controller.py:419-424), both call sequences.  usage: python tools/dev/fuzz_facade.py [n] [first_seed]"""
import copy
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from rotors_mpc_controller_amd.controller import PositionNMPC
from rotors_mpc_controller_amd.params import load_params
from rotors_mpc_controller_amd.reference import ReferenceGenerator
from tests.oracle_solver import OracleOcpSolver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
base = load_params()
bad = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(31000 + seed)
    p = copy.deepcopy(base)
    p["solver"]["horizon_steps"] = int(rng.choice([5, 10, 20, 30]))
    p["solver"]["dt"] = float(rng.choice([0.02, 0.05, 0.1]))
    for key, n_ in (("position_weight", 3), ("velocity_weight", 3), ("quaternion_weight", 4), ("rate_weight", 3), ("control_weight", 4)):
        p["solver"][key] = [float(v) for v in 10.0 ** rng.uniform(-1, 1.5, n_)]
    p["solver"]["terminal_weight"] = [float(v) for v in 10.0 ** rng.uniform(-0.5, 1.5, 13)]
    p["solver"]["regularization"] = float(rng.choice([1e-3, 7e-3, 0.05]))
    p["vehicle"]["mass"] = float(rng.uniform(0.4, 2.5))
    jx, jy, jz = rng.uniform(0.004, 0.03, 3)
    p["vehicle"]["inertia"] = [float(jx), 0.0, 0.0, 0.0, float(jy), 0.0, 0.0, 0.0, float(jz)]
    p["vehicle"]["arm_length"] = float(rng.uniform(0.1, 0.35))
    p["vehicle"]["rotor_moment_constant"] = float(rng.uniform(0.005, 0.03))
    one_call = bool(rng.integers(0, 2))
    g = PositionNMPC(p, one_call=one_call)
    def oracle_with_polish(cfg):
        so_ = OracleOcpSolver(cfg)
        so_.c.qp_polish = 1                # the same active-set policy as the GPU default path
        return so_
    o = PositionNMPC(p, solver_factory=oracle_with_polish)
    gen = ReferenceGenerator(p["reference"]); gen.update_hover_thrust(g.hover_thrust)
    state = dict(position=rng.normal(0, 0.6, 3) + [0, 0, 1.0], velocity=rng.normal(0, 0.6, 3),
                 quaternion=(lambda a, ax: np.concatenate([[np.cos(a / 2)], np.sin(a / 2) * ax / np.linalg.norm(ax)]))(rng.uniform(0, 0.5), rng.normal(size=3)) * rng.uniform(0.5, 2.0),
                 body_rates=rng.normal(0, 0.8, 3))
    worst = 0.0; sts = []
    for tick in range(4):
        gen.update_defaults(rng.normal(0, 0.5, 3) + [0, 0, 1.0], rng.normal(0, 0.2, 3), float(rng.uniform(-1, 1)))
        ref = gen.build_horizon(g.horizon, g.dt)
        ug, sg = g.solve(state, ref); uo, so = o.solve(state, ref)
        sts.append((sg, so))
        if sg == 0 and so == 0:
            worst = max(worst, float(np.abs(ug - uo).max()))
        # the plant moves a little between ticks
        state["position"] = state["position"] + 0.05 * state["velocity"]; state["velocity"] = state["velocity"] + rng.normal(0, 0.05, 3)
    flag = "" if (worst < 1e-8 and all(a == b for a, b in sts)) else "   <-- CHECK"
    bad += bool(flag)
    print(f"seed {seed:3d} N={g.horizon:2d} dt={g.dt} one_call={int(one_call)} statuses {sts} worst |du0| {worst:.1e}{flag}", flush=True)
    g.close() if hasattr(g, "close") else None
print("draws to check:", bad)
