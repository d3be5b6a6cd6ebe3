#!/usr/bin/env python3
"""Runs the UNMODIFIED reference controller (/root/reference/src/rotors_mpc_controller/controller.py)
on top of the Level-B shims (tests/levelb) with the CPU oracle as backend, and records what it
receives and returns over a short closed loop: tests/golden/levelb_closed_loop.npz.

What this pins: the reference's own staging code (quaternion normalisation, x0 pin, cold / unshifted
warm start, yref stacking, failure path; controller.py:385-463), its parameter derivation (:63-172)
and its OCP construction (:175-264) as seen by a solver.  What it does NOT pin: acados' arithmetic
(the backend is this repository's oracle) -- that stays "parity unpinned".
Only data is stored.  Dev container only (needs /root/reference)."""
import importlib.util
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "levelb"))
REF = Path("/root/reference/src/rotors_mpc_controller")


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    import os
    os.environ["ROTORS_MPC_PARAMS"] = "/root/reference/config/params.yaml"
    os.environ["LEVELB_BACKEND"] = "oracle"
    ctrl_mod = load("ref_controller", REF / "controller.py")
    ref_mod = load("ref_reference", REF / "reference.py")
    par_mod = load("ref_params", REF / "params.py")
    params = par_mod.load_params()
    params["solver"]["codegen_directory"] = tempfile.mkdtemp()
    ctrl = ctrl_mod.PositionNMPC(params)
    gen = ref_mod.ReferenceGenerator(params["reference"])
    gen.update_hover_thrust(ctrl.hover_thrust)
    from oracle import oracle as O
    oc = ctrl._solver._oc
    rng = np.random.default_rng(7)
    runs = []
    for trial in range(6):
        x = np.zeros(13); x[0:3] = [0, 0, 1] + rng.normal(0, 0.5, 3); x[3:6] = rng.normal(0, 0.5, 3)
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); a = rng.uniform(0, 0.4)
        x[6:10] = np.r_[np.cos(a / 2), np.sin(a / 2) * ax]; x[10:13] = rng.normal(0, 0.5, 3)
        if trial == 3:
            gen.set_target(position=np.array([1.0, -1.0, 2.0]), yaw=0.8)
        ctrl._prev_solution_valid = False
        states, cmds, stats = [], [], []
        for t in range(8):
            state = dict(position=x[0:3].copy(), velocity=x[3:6].copy(),
                         quaternion=x[6:10] * (1.0 + 0.01 * t), body_rates=x[10:13].copy())   # un-normalised on purpose
            u0, st = ctrl.solve(state, gen.build_horizon(ctrl.horizon, ctrl.dt))
            states.append(np.r_[state["position"], state["velocity"], state["quaternion"], state["body_rates"]])
            cmds.append(u0); stats.append(st)
            x = O.integrate(oc, np.r_[x[0:6], x[6:10] / np.linalg.norm(x[6:10]), x[10:13]], u0)[0]
        ref_h = gen.build_horizon(ctrl.horizon, ctrl.dt)
        runs.append(dict(states=np.array(states), cmds=np.array(cmds), status=np.array(stats),
                         positions=ref_h["positions"][0], yaw=ref_h["yaws"][0]))
    np.savez_compressed(Path(__file__).parent / "levelb_closed_loop.npz",
                        states=np.array([r["states"] for r in runs]), cmds=np.array([r["cmds"] for r in runs]),
                        status=np.array([r["status"] for r in runs]),
                        setpoints=np.array([r["positions"] for r in runs]), yaws=np.array([r["yaw"] for r in runs]),
                        hover_thrust=ctrl.hover_thrust, horizon=ctrl.horizon, dt=ctrl.dt,
                        lbu=ctrl.input_bounds[0], ubu=ctrl.input_bounds[1],
                        probed_mass=ctrl._solver._cfg["mass"], probed_gravity=ctrl._solver._cfg["gravity"])
    print("levelb_closed_loop.npz: 6 runs x 8 ticks through the reference's PositionNMPC.solve; statuses",
          np.unique(np.array([r["status"] for r in runs])), "probed mass", ctrl._solver._cfg["mass"],
          "g", ctrl._solver._cfg["gravity"], "J", ctrl._solver._cfg["J"])


if __name__ == "__main__":
    main()
