import sys
sys.path.insert(0,'/root/repo')
import numpy as np
from oracle import oracle as O
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from tests.fuzz_draws import draw, oracle_config
over, x0, yref, ye, hov, di, rng = draw(1910)
s = NmpcOcpSolver(_lib.default_config(**over))
c = oracle_config(over)
out = s.solve_batch(x0, yref, ye, want_traj=True)
it, ps = s.counts()
ref = O.solve_batch(c, x0, yref, ye, want_traj=True, nthreads=16)
bad = np.nonzero(out["status"] != ref["status"])[0]
print("mismatching instances", bad, "gpu status", out["status"][bad], "oracle status", ref["status"][bad], "gpu iters", it[bad], "oracle iters", ref["iters"][bad], "gpu passes", ps[bad], "oracle passes", ref["passes"][bad])
nz = np.nonzero(ref["status"] != 0)[0]
print("oracle status!=0:", nz, ref["status"][nz], "gpu there:", out["status"][nz], "iters", it[nz], ref["iters"][nz])
print("x0 of mismatching:", x0[bad])
