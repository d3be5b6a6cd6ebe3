#!/usr/bin/env python3
"""Dev check of the general FP64 kernels k_team_qp / k_team_qp_list (nmpc_team_as.hpp, MODE 1 / 2) against the oracle:
plain interior point, default split path, single-kernel path, per-stage linearisation, warm start, trajectories."""
import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from oracle import oracle as O
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, hover_reference, sample_x0
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)
WILD = dict(sigma_p=3.0, sigma_v=3.0, max_angle_deg=90.0, sigma_w=3.0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
worst = 0.0
def run(name, over, orc_over, dist, seed, N=20, warm=False, env=None):
    global worst
    for k, v in (env or {}).items(): os.environ[k] = v
    s = NmpcOcpSolver(_lib.default_config(N=N, max_batch=B, **over))
    for k in (env or {}): del os.environ[k]
    c = O.default_config(N=N, qp_gamma=0.0, **orc_over)
    yref, ye = hover_reference(N, 0.68 * 9.81 / 4)
    x0 = sample_x0(B, seed, **dist)
    o = s.solve_batch(x0, yref, ye, want_traj=True)
    r = O.solve_batch(c, x0, yref, ye, want_traj=True, nthreads=8)
    st = s.stats()
    it_g = s.iterations()
    def rep(tag, o, r, it_g):
        global worst
        ok = (o["status"] == 0) & (r["status"] == 0)
        du = np.abs(o["u0"] - r["u0"])[ok].max() if ok.any() else 0
        dx = np.abs(o["x"] - r["x"])[ok].max() if ok.any() else 0
        worst = max(worst, du)
        print(f"{name:34s} {tag}: status gpu {np.bincount(o['status'], minlength=5)} orc {np.bincount(r['status'], minlength=5)} mism {(o['status'] != r['status']).sum()} "
              f"iters equal {np.array_equal(it_g, r['iters'])} (gpu max {it_g.max()} orc max {r['iters'].max()}) |du0| {du:.1e} |dx| {dx:.1e} passes max {st['polish_max']} tail {st['n_tail']}", flush=True)
    rep("cold", o, r, it_g)
    if warm:
        o2 = s.solve_batch(x0, yref, ye, x_init=o["x"], u_init=o["u"], want_traj=True)
        r2 = O.solve_batch(c, x0, yref, ye, x_init=r["x"], u_init=r["u"], want_traj=True, nthreads=8)
        st = s.stats()
        rep("warm", o2, r2, s.iterations())
    g = s.guard_check()
    if g == 0: print(f"{name:34s} guard bands clean", flush=True)
    s.close()
SH = _lib.FLAG_TEAM_MAPPING | _lib.FLAG_SHARE_COLD_START
for dist, nm, seed in ((NEAR_HOVER, "near", 0), (AGGRESSIVE, "aggr", 1), (WILD, "wild", 2)):
    run(f"plain ipm shared {nm}", dict(qp_polish=0), dict(qp_polish=0), dist, seed)
    run(f"plain ipm per-stage {nm}", dict(qp_polish=0, flags=_lib.FLAG_TEAM_MAPPING), dict(qp_polish=0), dist, seed, warm=True)
    run(f"default split {nm}", dict(), dict(qp_polish=1), dist, seed, warm=True)
    run(f"single kernel {nm}", dict(), dict(qp_polish=1), dist, seed, warm=True, env={"NMPC_TEAM_SPLIT": "0"})
    run(f"tight passes (3/6) {nm}", dict(qp_polish_passes=3, qp_polish_budget=6), dict(qp_polish=1, qp_polish_passes=3, qp_polish_budget=6), dist, seed, warm=True)
run("plain ipm N=5 wild", dict(qp_polish=0), dict(qp_polish=0), WILD, 3, N=5)
run("plain ipm N=57 wild", dict(qp_polish=0), dict(qp_polish=0), WILD, 4, N=57)
run("default N=57 wild", dict(), dict(qp_polish=1), WILD, 4, N=57, warm=True)
run("attempt after ipm (pol_mu 1e-2)", dict(qp_polish_mu=1e-2), dict(qp_polish=1, qp_polish_mu=1e-2), AGGRESSIVE, 5)
run("3 integrator steps, default", dict(sim_num_steps=3), dict(qp_polish=1, sim_num_steps=3), WILD, 7, warm=True)
run("4 integrator steps, plain ipm", dict(sim_num_steps=4, qp_polish=0, flags=_lib.FLAG_TEAM_MAPPING), dict(qp_polish=0, sim_num_steps=4), AGGRESSIVE, 8, warm=True)
run("3 integrator steps, shared N=57", dict(sim_num_steps=3), dict(qp_polish=1, sim_num_steps=3), AGGRESSIVE, 9, N=57, warm=True)
run("iter cap 3, reported", dict(qp_polish=0, qp_iter_max=3, qp_maxiter_status=2), dict(qp_polish=0, qp_iter_max=3, qp_maxiter_status=2), NEAR_HOVER, 6)
run("iter cap 3, tolerated", dict(qp_polish=0, qp_iter_max=3), dict(qp_polish=0, qp_iter_max=3), NEAR_HOVER, 6)
print("worst |du0| over all rows: %.2e" % worst)
