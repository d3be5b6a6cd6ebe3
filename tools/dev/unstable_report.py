#!/usr/bin/env python3
"""Dev: the unstable-plant test of tests/test_gpu_parity.py in numbers: statuses and |du0| by the way an instance ended."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from oracle import oracle as O
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import sample_x0
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
from tests.fuzz_draws import WILD, oracle_config
GM = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
for N in (31, 120):
    for polish in (1, 0):
        over = dict(N=N, dt=0.1, mass=0.4738978976479069, inertia=[0.0017, 0.006, 0.012],
                    rotor_x=[0.4541, 0.0, -0.4541, 0.0], rotor_y=[0.0, 0.4541, 0.0, -0.4541], rotor_z=[-0.0141, 0.0141, -0.0141, 0.0141],
                    lbu=[0.0452] * 4, ubu=[2.0395] * 4,
                    W=[0.1868, 4.7625, 0.1212, 70.9006, 3.7842, 57.0244, 0.0357, 32.6734, 12.5504, 0.1446, 29.7859, 5.8384, 0.409,
                       11.6702, 0.8986, 3.7607, 0.0134],
                    W_e=[0.127, 13.7797, 13.6895, 2.7246, 3.6848, 4.8743, 0.2726, 140.7136, 0.5401, 23.0029, 17.7842, 0.149, 23.2557],
                    levenberg_marquardt=0.0, sim_num_steps=1, lm_scaled_by_dt=1, cost_scaled_by_dt=1,
                    flags=_lib.FLAG_TEAM_MAPPING, max_batch=512, qp_polish=polish, qp_growth_max=GM)
        s = NmpcOcpSolver(_lib.default_config(**over))
        c = oracle_config(s.config)
        x0 = sample_x0(511, 9021, **WILD)
        hov = over["mass"] * 9.81 / 4.0
        yref = np.zeros((N, 17)); yref[:, 2] = 1.0; yref[:, 6] = 1.0; yref[:, 13:] = hov
        ye = yref[0, :13].copy()
        out = s.solve_batch(x0, yref, ye); it, ps = s.counts()
        ref = O.solve_batch(c, x0, yref, ye, nthreads=8)
        both = (ref["status"] == 0) & (out["status"] == 0)
        acc = both & (ps > 0) & (ref["passes"] > 0)
        ipm = both & ~acc
        du = np.abs(out["u0"] - ref["u0"]).max(1)
        print(f"N={N} polish={polish}: status gpu {np.bincount(out['status'], minlength=5)} orc {np.bincount(ref['status'], minlength=5)} mismatches {(out['status'] != ref['status']).sum()} "
              f"iters equal {(it == ref['iters']).mean():.3f}  accepted-AS {acc.sum()} |du0| {du[acc].max() if acc.any() else 0:.1e}  ipm-ended {ipm.sum()} |du0| {du[ipm].max() if ipm.any() else 0:.1e}"
              f"  max growth orc {ref['growth'].max():.1e}", flush=True)
        mm = np.where(out["status"] != ref["status"])[0]
        for i in mm[:6]: print(f"    inst {i}: gpu st {out['status'][i]} it {it[i]} ps {ps[i]} | orc st {ref['status'][i]} it {ref['iters'][i]} ps {ref['passes'][i]} growth {ref['growth'][i]:.2e}")
        w = np.argsort(-np.where(ipm, du, 0))[:4]
        for i in w:
            if ipm[i]: print(f"    ipm-ended inst {i}: |du0| {du[i]:.1e} gpu it {it[i]} ps {ps[i]} | orc it {ref['iters'][i]} ps {ref['passes'][i]} growth {ref['growth'][i]:.2e}")
        w = np.argsort(-np.where(acc, du, 0))[:5]
        for i in w:
            if acc[i]: print(f"    accepted inst {i}: |du0| {du[i]:.1e} gpu it {it[i]} ps {ps[i]} | orc it {ref['iters'][i]} ps {ref['passes'][i]} growth {ref['growth'][i]:.2e}")
        s.close()
