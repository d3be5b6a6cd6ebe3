#!/usr/bin/env python3
"""Diagnostic (NMPC_DEBUG_NAN prof build): first non-finite value in the general kernel's factor sweep for one instance of
tools/dev/unstable_n120.py.  usage: nan_probe2.py <N> <inst>"""
import sys, os, ctypes as C
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
os.environ["ROTORS_NMPC_LIB"] = str(ROOT / "rotors_mpc_controller_amd" / "librotors_nmpc_hip_prof.so")
import runpy
N, inst = int(sys.argv[1]), int(sys.argv[2])
sys.argv = [sys.argv[0], str(N)]
g = runpy.run_path(str(Path(__file__).resolve().parent / "unstable_n120.py"), run_name="x")
_lib, NmpcOcpSolver, over = g["_lib"], g["NmpcOcpSolver"], g["over"]
lib = _lib.load()
lib.nmpc_debug_prof_copy.argtypes = [C.c_void_p, C.c_void_p]; lib.nmpc_debug_prof_copy.restype = C.c_int
x0, yref, ye = g["x0"][inst:inst + 1], g["yref"], g["ye"]
s = NmpcOcpSolver(_lib.default_config(**dict(over, max_batch=4, qp_polish=0)))
o = s.solve_batch(x0, yref, ye)
host = np.zeros((8, 64), dtype=np.int64)
assert lib.nmpc_debug_prof_copy(s._h, host.ctypes.data) == 64
v = -int(host[7, 0])
print("status", o["status"], "ipm", s.stats()["iter_max"], "raw", host[7, 0], "-> iteration", (v - 1000000) // 10000, "stage", ((v - 1000000) % 10000) // 10, "code", v % 10,
      "(1 stage tiles, 4 barrier diagonal / rhat, 2 P in, 3 Huu)")
