#!/usr/bin/env python3
"""Details of one draw of tools/dev/fuzz_parity.py: usage: python tools/dev/fuzz_one.py <seed> [key=value overrides]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from oracle import oracle as O
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
from tests.fuzz_draws import draw, oracle_config
seed = int(sys.argv[1])
over, x0, yref, ye, hov, di, rng = draw(seed)
for kv in [a for a in sys.argv[2:] if "=" in a]:
    k, v = kv.split("=")
    over[k] = type(over[k])(float(v)) if k in over and not isinstance(over[k], list) else float(v)
N, B = over["N"], over["max_batch"]
print({k: (v if not isinstance(v, list) else [round(x, 4) for x in v]) for k, v in over.items()})
for polish in (1, 0):
    s = NmpcOcpSolver(_lib.default_config(**dict(over, qp_polish=polish)))
    c = oracle_config(s.config)
    out = s.solve_batch(x0, yref, ye, want_traj=True); it, ps = s.counts()
    ref = O.solve_batch(c, x0, yref, ye, want_traj=True, nthreads=16)
    print(f"--- qp_polish={polish}: gpu status hist {np.bincount(out['status'], minlength=5)} oracle {np.bincount(ref['status'], minlength=5)}")
    d = np.abs(out["u0"] - ref["u0"]).max(1)
    idx = list(np.where(out["status"] != ref["status"])[0][:8]) + list(np.argsort(-d)[:6])
    for i in idx:
        print(f"inst {i}: gpu status {out['status'][i]} it {it[i]} passes {ps[i]} | oracle status {ref['status'][i]} it {ref['iters'][i]} passes {ref['passes'][i]} growth {ref['growth'][i]:.2e}"
              f" |du0| {d[i]:.2e} u0 gpu {np.round(out['u0'][i], 5)} oracle {np.round(ref['u0'][i], 5)}")
    st = s.stats(); print({k: st[k] for k in ("iter_mean", "iter_max", "polish_mean", "polish_max", "n_status", "n_tail")})
    if "--warm" in sys.argv:       # second solve, both sides from the oracle's first result
        out2 = s.solve_batch(x0, yref, ye, x_init=ref["x"], u_init=ref["u"], want_traj=True); it2, ps2 = s.counts()
        ref2 = O.solve_batch(c, x0, yref, ye, x_init=ref["x"], u_init=ref["u"], want_traj=True, nthreads=16)
        print(f"    warm: gpu status hist {np.bincount(out2['status'], minlength=5)} oracle {np.bincount(ref2['status'], minlength=5)}")
        for i in np.where(out2["status"] != ref2["status"])[0][:8]:
            print(f"    inst {i}: gpu status {out2['status'][i]} it {it2[i]} passes {ps2[i]} | oracle status {ref2['status'][i]} it {ref2['iters'][i]} passes {ref2['passes'][i]} growth {ref2['growth'][i]:.3e}"
                  f" max|x_init| {np.abs(ref['x'][i]).max():.2e}")
    s.close()
