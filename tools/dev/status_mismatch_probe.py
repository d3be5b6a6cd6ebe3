#!/usr/bin/env python3
"""For fuzz draws whose statuses differ from the oracle's on some instance: which instance, both statuses, iteration and pass counts,
cold and warm-started - run under different builds / schedules (environment) to tell an arithmetic edge from a build.
usage: python tools/dev/status_mismatch_probe.py <seed> [<seed> ...]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from oracle import oracle as O  # noqa: E402
from rotors_mpc_controller_amd import _lib  # noqa: E402
from rotors_mpc_controller_amd.solver import NmpcOcpSolver  # noqa: E402
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
from tests.fuzz_draws import draw, oracle_config  # noqa: E402

for seed in (int(a) for a in sys.argv[1:]):
    over, x0, yref, ye, hov, di, rng = draw(seed)
    s = NmpcOcpSolver(_lib.default_config(**over))
    c = oracle_config(over)
    out = s.solve_batch(x0, yref, ye, want_traj=True)
    it, ps = s.counts()
    ref = O.solve_batch(c, x0, yref, ye, want_traj=True, nthreads=16)
    out2 = s.solve_batch(x0, yref, ye, x_init=ref["x"], u_init=ref["u"], want_traj=True)
    it2, ps2 = s.counts()
    ref2 = O.solve_batch(c, x0, yref, ye, x_init=ref["x"], u_init=ref["u"], want_traj=True, nthreads=16)
    for tag, o, r, i, p in (("cold", out, ref, it, ps), ("warm", out2, ref2, it2, ps2)):
        bad = np.nonzero(o["status"] != r["status"])[0]
        for b in bad:
            print(f"seed {seed} {tag} instance {b}: status gpu {o['status'][b]} oracle {r['status'][b]} | iterations {i[b]} / {r['iters'][b]} | passes {p[b]} / {r['passes'][b]} "
                  f"| max|x0| {np.abs(x0[b]).max():.2f} | max|x_init| {np.abs(ref['x'][b]).max():.3g}")
    s.close()
