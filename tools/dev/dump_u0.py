import sys, ctypes, numpy as np
sys.path.insert(0, '.')
import torch
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import NEAR_HOVER, hover_reference, sample_x0
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
N, B, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
s = NmpcOcpSolver(_lib.default_config(N=N, max_batch=B))
yref, ye = hover_reference(N, 0.68 * 9.81 / 4)
x0 = sample_x0(B, 5, **NEAR_HOVER)
o = s.solve_batch(x0, yref, ye, want_traj=True)
lib = _lib.load(); lib.nmpc_device_passes.restype = ctypes.c_void_p
hip = ctypes.CDLL("libamdhip64.so"); buf = np.zeros(B, np.int32)
hip.hipMemcpy(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(lib.nmpc_device_passes(s._h)), B * 4, 2)
np.savez(out, u0=o["u0"], u=o["u"], x=o["x"], passes=buf)
