#!/usr/bin/env python3
"""Long-horizon solves through the block-parallel tail (default from N = 160) against the sequential work list (NMPC_BLOCK_TAIL=0) and,
on a sample, against the CPU oracle: horizons x distributions x seeds, cold and warm-started.  GPU box, a few minutes.
usage: python tools/dev/tail_sweep_check.py [B]"""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from oracle import oracle as O  # noqa: E402
from rotors_mpc_controller_amd import _lib  # noqa: E402
from rotors_mpc_controller_amd.solver import NmpcOcpSolver  # noqa: E402
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, hover_reference, sample_x0  # noqa: E402
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)

WILD = dict(sigma_p=3.0, sigma_v=3.0, max_angle_deg=90.0, sigma_w=3.0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bad = 0
for N in (160, 200, 300, 450, 600):
    yref, ye = hover_reference(N, 0.68 * 9.81 / 4.0)
    for name, dist in (("near", NEAR_HOVER), ("aggr", AGGRESSIVE), ("wild", WILD)):
        for seed in (1, 2):
            x0 = sample_x0(B, 100 * N + seed, **dist)
            res = []
            for tail in ("0", "1"):
                os.environ["NMPC_BLOCK_TAIL"] = tail
                s = NmpcOcpSolver(_lib.default_config(N=N, max_batch=B))
                a = s.solve_batch(x0, yref, ye, want_traj=True)
                ca = s.counts()
                ok = a["status"] == 0
                xi = np.where(ok[:, None, None], a["x"], np.tile(x0[:, None, :], (1, N + 1, 1)))
                ui = np.where(ok[:, None, None], a["u"], 0.0)
                res.append((a, ca, s.tail_states(B)))
                if tail == "0":
                    warm_in = (xi, ui)
                w = s.solve_batch(x0, yref, ye, x_init=warm_in[0], u_init=warm_in[1])
                res[-1] = res[-1] + (w, s.counts())
                s.close()
            (a, ca, _, wa, cwa), (b, cb, (blocks, states), wb, cwb) = res
            ok = a["status"] == 0
            scale = np.maximum(1.0, np.abs(a["x"][ok]).max()) if ok.any() else 1.0
            line = dict(st=bool(np.array_equal(a["status"], b["status"])), it=bool(np.array_equal(ca[0], cb[0])), ps=bool(np.array_equal(ca[1], cb[1])),
                        du0=float(np.abs(a["u0"][ok] - b["u0"][ok]).max()) if ok.any() else 0.0,
                        dx=float(np.abs(a["x"][ok] - b["x"][ok]).max() / scale) if ok.any() else 0.0,
                        wst=bool(np.array_equal(wa["status"], wb["status"])), wps=bool(np.array_equal(cwa[1], cwb[1])),
                        wdu0=float(np.abs(wa["u0"] - wb["u0"])[(wa["status"] == 0)].max()) if (wa["status"] == 0).any() else 0.0)
            # oracle on 6 instances
            c = O.default_config(N=N, qp_gamma=0.0, qp_polish=1)
            idx = np.arange(0, B, B // 6)[:6]
            ref = O.solve_batch(c, x0[idx], yref, ye, nthreads=6)
            line["oracle_st"] = bool(np.array_equal(ref["status"], b["status"][idx]))
            oko = (ref["status"] == 0) & (b["status"][idx] == 0)
            line["oracle_du0"] = float(np.abs(ref["u0"][oko] - b["u0"][idx][oko]).max()) if oko.any() else 0.0
            good = line["st"] and line["it"] and line["ps"] and line["wst"] and line["wps"] and line["du0"] < 1e-8 and line["wdu0"] < 1e-8 and line["dx"] < 1e-8 \
                and line["oracle_st"] and line["oracle_du0"] < 1e-7
            bad += not good
            print(f"N {N:3d} {name} seed {seed}: blocks {blocks} tail finished {int((states == 3).sum())} fallback {int((states == 5).sum())} status!=0 {int((a['status'] != 0).sum())} "
                  f"passes max {int(np.abs(ca[1]).max())} ipm max {int(ca[0].max())} | {line}{'' if good else '   <-- CHECK'}", flush=True)
print("to check:", bad)
