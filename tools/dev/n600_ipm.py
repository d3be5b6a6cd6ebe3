import sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle as O
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import NEAR_HOVER, hover_reference, sample_x0
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
N = int(sys.argv[1]); B = 64
s = NmpcOcpSolver(_lib.default_config(N=N, max_batch=B, qp_polish=0))
yref, ye = hover_reference(N, 0.68 * 9.81 / 4)
x0 = sample_x0(1024, 5, **NEAR_HOVER)[:B]
o = s.solve_batch(x0, yref, ye, want_traj=True)
r = O.solve_batch(O.default_config(N=N, qp_gamma=0.0, qp_polish=0), x0, yref, ye, want_traj=True, nthreads=8)
print("N", N, "gpu stats", {k: v for k, v in s.stats().items() if k in ("iter_mean", "iter_max", "n_status")}, "oracle iters", r["iters"][:16])
print("max |u0 - oracle|", np.abs(o["u0"] - r["u0"]).max(), "max |u - oracle|", np.abs(o["u"] - r["u"]).max())
