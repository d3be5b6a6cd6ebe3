#!/usr/bin/env python3
"""Dev: does the plain-IPM rate depend on what ran before in the process?"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import NEAR_HOVER, hover_reference, sample_x0
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
B, N = 4096, 20
dev = torch.device("cuda", 0)
x0 = torch.from_numpy(sample_x0(B, 0, **NEAR_HOVER)).to(dev)
yr, ye = hover_reference(N, 0.68 * 9.81 / 4)
yref = torch.from_numpy(np.tile(yr, (B, 1, 1))).to(dev).contiguous(); yref_e = torch.from_numpy(np.tile(ye, (B, 1))).to(dev).contiguous()
u0 = torch.zeros(B, 4, dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev)
def rate(sv, steps=50, warm=5):
    sv.set_timing(False)
    f = lambda: sv.solve_batch_device(B, x0.data_ptr(), yref.data_ptr(), yref_e.data_ptr(), False, u0.data_ptr(), status_ptr=st.data_ptr(), stream=stream.cuda_stream)
    for _ in range(warm): f()
    torch.cuda.synchronize(dev); t = time.perf_counter()
    for _ in range(steps): f()
    torch.cuda.synchronize(dev); return 1e3 * (time.perf_counter() - t) / steps
mk = lambda **o: NmpcOcpSolver(_lib.default_config(N=N, max_batch=B, **o))
a = mk(qp_polish=0); print("plain ipm alone        %.4f ms" % rate(a)); print("again                  %.4f ms" % rate(a))
b = mk(); print("default                %.4f ms" % rate(b, 200, 20))
print("plain ipm after default %.4f ms" % rate(a))
c = mk(qp_polish=0); print("NEW plain ipm solver   %.4f ms" % rate(c))
d = mk(flags=_lib.FLAG_TEAM_MAPPING); print("no-share               %.4f ms" % rate(d))
e = mk(qp_polish=0); print("NEW plain ipm solver 2 %.4f ms" % rate(e)); print("old one again          %.4f ms" % rate(a))
a.set_timing(True); a.solve_batch_device(B, x0.data_ptr(), yref.data_ptr(), yref_e.data_ptr(), False, u0.data_ptr(), status_ptr=st.data_ptr(), stream=stream.cuda_stream); print(a.stats())
e.set_timing(True); e.solve_batch_device(B, x0.data_ptr(), yref.data_ptr(), yref_e.data_ptr(), False, u0.data_ptr(), status_ptr=st.data_ptr(), stream=stream.cuda_stream); print(e.stats())
