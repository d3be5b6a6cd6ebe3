"""dev: permutation equivariance at N=600 with per-instance pass / iteration counts."""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, '.')
import torch
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import NEAR_HOVER, hover_reference, sample_x0
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
N, B = int(sys.argv[1]), int(sys.argv[2])
s = NmpcOcpSolver(_lib.default_config(N=N, max_batch=B))
yref, ye = hover_reference(N, 0.68 * 9.81 / 4)
x0 = sample_x0(B, 5, **NEAR_HOVER)
def run(x):
    o = s.solve_batch(x, yref, ye)
    it = torch.empty(B, dtype=torch.int32, device="cuda")
    C.cdll.LoadLibrary  # noqa
    lib = _lib.load()
    # iterations live on the device
    import ctypes
    lib.nmpc_device_iterations.restype = ctypes.c_void_p
    p = lib.nmpc_device_iterations(s._h)
    hip = ctypes.CDLL("libamdhip64.so")
    buf = np.zeros(B, np.int32)
    hip.hipMemcpy(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p), B * 4, 2)
    lib.nmpc_device_passes.restype = ctypes.c_void_p
    buf2 = np.zeros(B, np.int32)
    hip.hipMemcpy(buf2.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(lib.nmpc_device_passes(s._h)), B * 4, 2)
    return o["u0"], np.stack([buf, buf2], 1)
u_a, it_a = run(x0)
perm = np.random.default_rng(5).permutation(B)
u_b, it_b = run(x0[perm])
d = np.abs(u_b - u_a[perm]).max(axis=1)
bad = np.where(d > 0)[0]
print("mismatches", len(bad))
for i in bad[:20]:
    print("perm idx", i, "orig", perm[i], "diff", d[i], "(iters, passes) a/b", it_a[perm[i]], it_b[i], "wave-mates a", it_a[perm[i] // 4 * 4: perm[i] // 4 * 4 + 4].tolist(), "b", it_b[i // 4 * 4: i // 4 * 4 + 4].tolist())
u_c, it_c = run(x0)
print("same batch twice: max diff", np.abs(u_c - u_a).max())
