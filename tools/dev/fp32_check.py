"""FP32 end-to-end accuracy of the default (team + polish) path against the FP64 oracle."""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle as O
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, hover_reference, sample_x0
yref, ye = hover_reference(20, 0.68 * 9.81 / 4)
s = NmpcOcpSolver(_lib.default_config(max_batch=1024, dtype=_lib.DTYPE_F32))
for name, dist in (("near", NEAR_HOVER), ("aggr", AGGRESSIVE), ("wild", dict(sigma_p=3.0, sigma_v=3.0, max_angle_deg=90.0, sigma_w=3.0))):
    x0 = sample_x0(1000, 4, **dist)
    out = s.solve_batch(x0, yref, ye)
    ref = O.solve_batch(O.default_config(qp_polish=1), x0, yref, ye)
    st = s.stats()
    ok = out["status"] == 0
    e = np.abs(out["u0"] - ref["u0"])[ok]
    print(name, "status", np.bincount(out["status"]), "max|u0 err| %.2e median %.2e" % (e.max(), np.median(e)),
          "ipm iters %.2f passes %.2f accepted %d" % (st["iter_mean"], st["polish_mean"], st["n_polished"]))
