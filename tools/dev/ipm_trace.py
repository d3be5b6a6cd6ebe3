#!/usr/bin/env python3
"""Diagnostic: the interior-point iterates of one instance of tools/dev/unstable_n120.py, tile form against row form, by
capping qp_iter_max at k = 1, 2, ... (status 2 hands back the current iterate).  usage: ipm_trace.py <N> <inst> [kmax]"""
import sys, os
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import runpy
N, inst = int(sys.argv[1]), int(sys.argv[2]); kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 45
sys.argv = [sys.argv[0], "3"]          # a tiny horizon for the module-level run of unstable_n120 (only its config is wanted)
g = runpy.run_path(str(Path(__file__).resolve().parent / "unstable_n120.py"), run_name="x")
_lib, NmpcOcpSolver, sample_x0, WILD = g["_lib"], g["NmpcOcpSolver"], g["sample_x0"], g["WILD"]
over = dict(g["over"], N=N, max_batch=4, qp_polish=0)
x0 = sample_x0(511, 9021, **WILD)[inst:inst + 1]
hov = over["mass"] * 9.81 / 4.0
yref = np.zeros((N, 17)); yref[:, 2] = 1.0; yref[:, 6] = 1.0; yref[:, 13:] = hov
ye = yref[0, :13].copy()
for k in list(range(1, kmax + 1)):
    res = {}
    for name, env in (("tile", {}), ("row", {"NMPC_TEAM_MFMA": "0"})):
        for kk, v in env.items():
            os.environ[kk] = v
        s = NmpcOcpSolver(_lib.default_config(**dict(over, qp_iter_max=k)))
        o = s.solve_batch(x0, yref, ye, want_traj=True)
        res[name] = (int(o["status"][0]), o["u"][0].copy(), s.stats()["iter_max"])
        s.close()
        for kk in env:
            del os.environ[kk]
    du = np.abs(res["tile"][1] - res["row"][1]).max()
    print(f"cap {k:2d}: tile status {res['tile'][0]} it {res['tile'][2]} | row status {res['row'][0]} it {res['row'][2]} | max |u_tile - u_row| {du:.2e} max|u| {np.abs(res['row'][1]).max():.3g}", flush=True)
