#!/usr/bin/env python3
"""Generates the committed golden fixtures.  Run from the repo root: python tests/golden/make_golden.py

1. rti_cold_start.npz -- inputs and expected outputs of one cold-start SQP-RTI per instance,
   produced by the CPU oracle (oracle/nmpc_oracle.c).  NOT acados output: acados is not
   available in this environment ("parity unpinned").
2. reference_horizon.json / params_coerced.json -- outputs of the two reference modules that DO
   import standalone here (reference.py, params.py; loaded by file path from /root/reference,
   numpy/PyYAML only).  They pin this package's own ReferenceGenerator / load_params.
   Only data is stored, never reference source text.
"""
import importlib.util
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")


def golden_rti():
    from oracle import oracle as O
    from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, hover_reference, sample_x0
    c = O.default_config(qp_gamma=0.0)
    x0 = np.concatenate([sample_x0(24, 0, **NEAR_HOVER), sample_x0(24, 1, **AGGRESSIVE),
                         sample_x0(16, 2, sigma_p=3.0, sigma_v=3.0, max_angle_deg=90.0, sigma_w=3.0)])
    yref, ye = hover_reference(c.N, c.mass * c.gravity / 4.0)
    out = O.solve_batch(c, x0, yref, ye, want_traj=True)
    # the same instances with the active-set polish (exact QP solutions; the team kernel's default)
    outp = O.solve_batch(O.default_config(qp_gamma=0.0, qp_polish=1), x0, yref, ye, want_traj=True)
    np.savez_compressed(HERE / "rti_cold_start.npz", x0=x0, yref=yref, yref_e=ye, u0=out["u0"],
                        status=out["status"], iters=out["iters"], x=out["x"], u=out["u"],
                        u0_polish=outp["u0"], x_polish=outp["x"], u_polish=outp["u"], iters_polish=outp["iters"])
    print("rti_cold_start.npz:", x0.shape[0], "instances, iters", out["iters"].min(), "..", out["iters"].max())


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def golden_reference_modules():
    if not REF.exists():
        print("no /root/reference here: keeping the committed reference_horizon.json / params_coerced.json")
        return
    refmod = _load("ref_reference", REF / "src/rotors_mpc_controller/reference.py")
    parmod = _load("ref_params", REF / "src/rotors_mpc_controller/params.py")
    os.environ["ROTORS_MPC_PARAMS"] = str(REF / "config/params.yaml")
    params = parmod.load_params()
    params.pop("params_yaml", None)
    (HERE / "params_coerced.json").write_text(json.dumps(params, indent=1, sort_keys=True))
    cases = []
    gen = refmod.ReferenceGenerator(params["reference"])
    gen.update_hover_thrust(0.68 * 9.81 / 4.0)
    for label, setup in (("default_hover", None),
                         ("target_yaw", dict(position=[1.0, -2.0, 3.0], yaw=0.7)),
                         ("target_quat_rates", dict(position=[0.5, 0.5, 2.0], velocity=[0.1, 0.2, 0.3],
                                                    quaternion=[2.0, 0.0, 0.0, 2.0], body_rates=[0.1, 0.0, -0.1],
                                                    thrust=[1.0, 2.0, 3.0, 4.0]))):
        if setup:
            gen.set_target(**{k: (np.asarray(v, float) if k != "yaw" else v) for k, v in setup.items()})
        hz = gen.build_horizon(20, 0.05)
        cases.append(dict(label=label, setup=setup, horizon=20, dt=0.05,
                          out={k: np.asarray(v).tolist() for k, v in hz.items()}))
    (HERE / "reference_horizon.json").write_text(json.dumps(cases))
    print("reference_horizon.json:", [c["label"] for c in cases])


if __name__ == "__main__":
    golden_rti()
    golden_reference_modules()
