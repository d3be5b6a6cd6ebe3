#!/usr/bin/env python3
"""sweep_workgroups.py -- emulate many workgroups of one kernel build in parallel processes; every access checked, u0 / status / iteration
counts against the oracle.  usage: sweep_workgroups.py <file.s> <kernel> <first> <last> [run_team_kernel options]"""
import sys, json
from pathlib import Path
from multiprocessing import Pool
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))

def one(args):
    import run_team_kernel as R
    sfile, kernel, wg, kw = args
    r = R.emulate(sfile, kernel, wg=wg, verbose=False, **kw)
    return dict(wg=wg, n=r["instructions"], err=r["error"], viol=[(v.kind, v.line, v.text, v.lane, hex(v.addr), v.note) for v in r["violations"][:5]],
                nviol=len(r["violations"]), u0=r["u0"].tolist(), status=r["status"].tolist(), iters=r["iters"].tolist(), inst=r["inst"], tpw=r["tpw"])

def main():
    sfile, kernel, a, b = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    kw = {}
    it = iter(sys.argv[5:])
    for k in it:
        k = k.lstrip("-")
        if k == "warm": kw["warm"] = True; continue
        v = next(it)
        kw[{"headers": "csrc", "include": "inc", "batch": "B"}.get(k, k)] = int(v) if v.lstrip("-").isdigit() else v
    with Pool(8) as p:
        res = p.map(one, [(sfile, kernel, wg, kw) for wg in range(a, b)], chunksize=1)
    from oracle import oracle as O
    from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, hover_reference, sample_x0
    B = kw.get("B", 256); steps = kw.get("steps", 4); polish = kw.get("polish", 0); seed = kw.get("seed", 8)
    c = O.default_config(N=20, qp_gamma=0.0, qp_polish=polish, sim_num_steps=steps)
    yref, ye = O.hover_yref(c)
    x0 = sample_x0(B, seed, **(AGGRESSIVE if kw.get("dist", "aggressive") == "aggressive" else NEAR_HOVER))
    xi = ui = None
    if kw.get("warm"):
        r0 = O.solve_batch(c, x0, yref, ye, want_traj=True, nthreads=8); xi, ui = r0["x"], r0["u"]
    ref = O.solve_batch(c, x0, yref, ye, x_init=xi, u_init=ui, nthreads=8)
    worst = 0.0; bad = 0; nv = 0; ninstr = 0; it_mis = 0
    for r in res:
        sl = slice(r["inst"], r["inst"] + r["tpw"])
        du = float(np.abs(np.array(r["u0"]) - ref["u0"][sl]).max())
        worst = max(worst, du); nv += r["nviol"]; ninstr += r["n"]
        it_mis += int((np.array(r["iters"]) != ref["iters"][sl]).sum())
        if r["err"] or r["nviol"] or du > 1e-9 or (np.array(r["status"]) != ref["status"][sl]).any():
            bad += 1
            print("workgroup", r["wg"], "error", r["err"], "violations", r["nviol"], r["viol"], "|du0|", du, "status", r["status"], ref["status"][sl])
    print(f"{Path(sfile).name} [{kernel}] workgroups {a}..{b - 1}: {ninstr} instructions emulated, {nv} access violations, {bad} workgroups to look at, "
          f"worst |u0 - oracle| {worst:.2e}, iteration-count mismatches {it_mis}")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
