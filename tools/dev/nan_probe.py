#!/usr/bin/env python3
"""Diagnostic (NMPC_DEBUG_NAN prof build): where does the first NaN of an instance of a fuzz draw appear?"""
import sys, os, ctypes as C
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
os.environ["ROTORS_NMPC_LIB"] = str(ROOT / "rotors_mpc_controller_amd" / "librotors_nmpc_hip_prof.so")
import runpy
seed = int(sys.argv[1]); insts = [int(v) for v in sys.argv[2:]]
sys.argv = [sys.argv[0], str(seed)]
g = runpy.run_path(str(Path(__file__).resolve().parent / "fuzz_one.py"), run_name="fuzz")
_lib, NmpcOcpSolver, over = g["_lib"], g["NmpcOcpSolver"], g["over"]
lib = _lib.load()
lib.nmpc_debug_prof_copy.argtypes = [C.c_void_p, C.c_void_p]; lib.nmpc_debug_prof_copy.restype = C.c_int
for inst in insts:
    x0, yref, ye = g["x0"][inst:inst + 1], g["yref"], g["ye"]
    if yref.ndim == 3:
        yref, ye = yref[inst:inst + 1], ye[inst:inst + 1]
    s = NmpcOcpSolver(_lib.default_config(**dict(over, max_batch=4)))
    o = s.solve_batch(x0, yref, ye)
    Bp = 64
    host = np.zeros((8, Bp), dtype=np.int64)
    assert lib.nmpc_debug_prof_copy(s._h, host.ctypes.data) == Bp
    print("inst", inst, "status", o["status"], "code", host[5, 0], "(1e6 + pass*1e5 + stage*100 + {1: stage tiles, 2: P in, 3: Huu})")
    s.close()
