#!/usr/bin/env python3
"""Fuzz draws whose commands differ from the oracle's beyond the asked accuracy although both sides end status 0: which instances, by how much,
after how many iterations / passes on either side, with which growth figure - cold and warm-started.
usage: python tools/dev/du0_probe.py <seed> [--long] [threshold, default 1e-6]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from oracle import oracle as O  # noqa: E402
from rotors_mpc_controller_amd import _lib  # noqa: E402
from rotors_mpc_controller_amd.solver import NmpcOcpSolver  # noqa: E402
import tools.dev._banner  # noqa: F401,E402
from tests.fuzz_draws import draw, oracle_config  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
LONG = dict(horizons=[160, 200, 256, 320, 400, 600], max_batch=65) if "--long" in sys.argv else {}
seed = int(args[0])
thr = float(args[1]) if len(args) > 1 else 1e-6
over, x0, yref, ye, hov, di, rng = draw(seed, **LONG)
if "--lane" in sys.argv:
    over.pop("qp_polish_ckpt")
    over.update(qp_polish=0, flags=over["flags"] & 1, qp_growth_max=0.0, qp_tol_step=0.0)
    B = min(over["max_batch"], 130); over["max_batch"] = B; x0 = x0[:B]
    if yref.ndim == 3:
        yref, ye = yref[:B], ye[:B]
s = NmpcOcpSolver(_lib.default_config(**over))
c = oracle_config(s.config, qp_polish=0 if "--lane" in sys.argv else 1)
scale = max(1.0, hov)
out = s.solve_batch(x0, yref, ye, want_traj=True); it, ps = s.counts()
ref = O.solve_batch(c, x0, yref, ye, want_traj=True, nthreads=16)
out2 = s.solve_batch(x0, yref, ye, x_init=ref["x"], u_init=ref["u"], want_traj=True); it2, ps2 = s.counts()
ref2 = O.solve_batch(c, x0, yref, ye, x_init=ref["x"], u_init=ref["u"], want_traj=True, nthreads=16)
print({k: (v if not isinstance(v, list) else [round(x, 4) for x in v]) for k, v in over.items()})
for tag, o, r, i, p in (("cold", out, ref, it, ps), ("warm", out2, ref2, it2, ps2)):
    ok = (o["status"] == 0) & (r["status"] == 0)
    du = np.abs(o["u0"] - r["u0"]).max(1) / scale
    for b in np.nonzero(ok & (du > thr))[0]:
        print(f"seed {seed} {tag} instance {b}: |du0| {du[b]:.2e} | iterations gpu {i[b]} oracle {r['iters'][b]} | passes gpu {p[b]} oracle {r['passes'][b]} | "
              f"oracle growth {r['growth'][b]:.2e} | u0 gpu {np.round(o['u0'][b], 6)} oracle {np.round(r['u0'][b], 6)}")
