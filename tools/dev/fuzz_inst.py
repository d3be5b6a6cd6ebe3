#!/usr/bin/env python3
"""One instance of a fuzz draw through every GPU path: usage fuzz_inst.py <seed> <instance>"""
import sys, os
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
import runpy
seed, inst = int(sys.argv[1]), int(sys.argv[2])
sys.argv = [sys.argv[0], str(seed)]
g = runpy.run_path(str(Path(__file__).resolve().parent / "fuzz_one.py"), run_name="fuzz")
O, _lib, NmpcOcpSolver, over, c = g["O"], g["_lib"], g["NmpcOcpSolver"], g["over"], g["c"]
x0, yref, ye = g["x0"][inst:inst + 1], g["yref"], g["ye"]
if yref.ndim == 3:
    yref, ye = yref[inst:inst + 1], ye[inst:inst + 1]
print("x0", np.round(x0[0], 3))
c.qp_polish = 1
r = O.solve_batch(c, x0, yref, ye, want_traj=True)
print("oracle polish: status", r["status"], "iters", r["iters"], "u0", r["u0"][0])
c.qp_polish = 0
r0 = O.solve_batch(c, x0, yref, ye, want_traj=True)
print("oracle ipm   : status", r0["status"], "iters", r0["iters"], "u0", r0["u0"][0])
print("oracle trajectory: max|x|", np.abs(r0["x"]).max(), "max |omega|", np.abs(r0["x"][0, :, 10:]).max())
for name, ov, env in (("team default", {}, {}), ("team one kernel", {}, {"NMPC_TEAM_SPLIT": "0"}), ("team plain ipm", dict(qp_polish=0), {}),
                      ("lane plain ipm", dict(qp_polish=0, flags=over["flags"] & 1), {})):
    for k, v in env.items():
        os.environ[k] = v
    s = NmpcOcpSolver(_lib.default_config(**dict(over, max_batch=4, **ov)))
    o = s.solve_batch(x0, yref, ye, want_traj=True)
    st = s.stats()
    print(f"{name:18s}: status {o['status']} ipm {st['iter_max']} passes {st['polish_max']} u0 {o['u0'][0]}")
    for k in env:
        del os.environ[k]
    s.close()
