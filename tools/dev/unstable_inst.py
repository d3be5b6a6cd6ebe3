#!/usr/bin/env python3
"""Diagnostic: one instance of tools/dev/unstable_n120.py through every GPU path.  usage: unstable_inst.py <N> <inst>"""
import sys, os
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
import runpy
N, inst = int(sys.argv[1]), int(sys.argv[2])
sys.argv = [sys.argv[0], str(N)]
g = runpy.run_path(str(Path(__file__).resolve().parent / "unstable_n120.py"), run_name="x")
O, _lib, NmpcOcpSolver, over = g["O"], g["_lib"], g["NmpcOcpSolver"], g["over"]
x0, yref, ye = g["x0"][inst:inst + 1], g["yref"], g["ye"]
for name, ov, env in (("team default", {}, {}), ("team plain ipm", dict(qp_polish=0), {}),
                      ("lane plain ipm", dict(qp_polish=0, flags=0), {})):
    for k, v in env.items():
        os.environ[k] = v
    s = NmpcOcpSolver(_lib.default_config(**dict(over, max_batch=4, **ov)))
    o = s.solve_batch(x0, yref, ye)
    st = s.stats()
    print(f"{name:18s}: status {o['status']} ipm {st['iter_max']} passes {st['polish_max']} u0 {o['u0'][0]}")
    for k in env:
        del os.environ[k]
    s.close()
