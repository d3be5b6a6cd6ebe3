#!/usr/bin/env python3
"""Details of one draw of tools/dev/fuzz_parity.py: usage: python tools/dev/fuzz_one.py <seed> [key=value overrides]"""
import sys
from pathlib import Path
import numpy as np
import os
sys.path.insert(0, os.environ.get("NMPC_ROOT", str(Path(__file__).resolve().parent.parent.parent)))
from oracle import oracle as O
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, sample_x0
from tests.oracle_solver import OracleOcpSolver
WILD = dict(sigma_p=3.0, sigma_v=3.0, max_angle_deg=90.0, sigma_w=3.0)
seed = int(sys.argv[1])
rng = np.random.default_rng(7000 + seed)
N = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 20, 24, 31, 40, 57]))
mass = float(rng.uniform(0.3, 4.0)); arm = float(rng.uniform(0.08, 0.5)); km = float(rng.uniform(0.003, 0.04)); hov = mass * 9.81 / 4.0
B = int(rng.choice([1, 3, 4, 5, 63, 64, 65, 130, 257, 511]))
over = dict(N=N, dt=float(rng.choice([0.01, 0.02, 0.05, 0.08, 0.1])), mass=mass,
            inertia=[float(v) for v in rng.uniform(0.002, 0.04, 3) * mass],
            rotor_x=[arm, 0.0, -arm, 0.0], rotor_y=[0.0, arm, 0.0, -arm], rotor_z=[-km, km, -km, km],
            lbu=[float(hov * rng.uniform(0.0, 0.5))] * 4, ubu=[float(hov * rng.uniform(1.3, 4.0))] * 4,
            W=[float(v) for v in 10.0 ** rng.uniform(-2, 2, 17)], W_e=[float(v) for v in 10.0 ** rng.uniform(-1, 2.5, 13)],
            levenberg_marquardt=float(rng.choice([0.0, 1e-4, 7e-3, 0.1, 1.0])), sim_num_steps=int(rng.choice([1, 2, 2, 3])),
            lm_scaled_by_dt=int(rng.integers(0, 2)), cost_scaled_by_dt=int(rng.integers(0, 2)),
            flags=_lib.FLAG_TEAM_MAPPING | int(rng.integers(0, 2)), max_batch=B,
            qp_polish_ckpt=int(rng.choice([0, 1, 4, 12, 100])))
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    over[k] = type(over[k])(float(v)) if not isinstance(over[k], list) else over[k]
N = over["N"]
print({k: (v if not isinstance(v, list) else [round(x, 4) for x in v]) for k, v in over.items()})
s = NmpcOcpSolver(_lib.default_config(**over))
c = OracleOcpSolver(s.config).c
c.qp_polish = 1
dist = [NEAR_HOVER, AGGRESSIVE, WILD][int(rng.integers(0, 3))]
x0 = sample_x0(B, 9000 + seed, **dist)
per_inst = bool(rng.integers(0, 2))
goal = rng.normal(0.0, 1.0, (B, 3)) + np.array([0.0, 0.0, 1.0]); vel = rng.normal(0.0, 0.3, (B, 3))
yref = np.zeros((B, N, 17)); ye = np.zeros((B, 13))
for k in range(N + 1):
    row = np.zeros((B, 13)); row[:, 0:3] = goal + vel * (k * over["dt"]); row[:, 3:6] = vel; row[:, 6] = 1.0
    if k < N:
        yref[:, k, :13] = row; yref[:, k, 13:] = hov
    else:
        ye[:] = row
if not per_inst:
    yref, ye = yref[0], ye[0]
out = s.solve_batch(x0, yref, ye, want_traj=True)
ref = O.solve_batch(c, x0, yref, ye, want_traj=True, nthreads=16)
it_g = np.zeros(B, np.int32)
print("gpu status hist", np.bincount(out["status"], minlength=5), "oracle", np.bincount(ref["status"], minlength=5))
d = np.abs(out["u0"] - ref["u0"]).max(1)
idx = np.argsort(-d)[:12]
for i in idx:
    print(f"inst {i}: gpu status {out['status'][i]} oracle {ref['status'][i]} iters(oracle) {ref['iters'][i]} |du0| {d[i]:.2e} u0 gpu {np.round(out['u0'][i], 4)} oracle {np.round(ref['u0'][i], 4)}")
st = s.stats(); print({k: st[k] for k in ("iter_mean", "iter_max", "polish_mean", "polish_max", "n_status")})
# the plain interior point on both sides, same instances
s2 = NmpcOcpSolver(_lib.default_config(**dict(over, qp_polish=0)))
c.qp_polish = 0
o2 = s2.solve_batch(x0, yref, ye); r2 = O.solve_batch(c, x0, yref, ye, nthreads=16)
print("plain IPM: gpu", np.bincount(o2["status"], minlength=5), "oracle", np.bincount(r2["status"], minlength=5), "max |du0| both ok",
      np.abs(o2["u0"] - r2["u0"])[(o2["status"] == 0) & (r2["status"] == 0)].max(), "gpu-polish vs gpu-ipm", np.abs(o2["u0"] - out["u0"])[(o2["status"] == 0) & (out["status"] == 0)].max())


def qp_check(c, x0i, yr, ye_, xo, uo):
    N = c.N
    xl = np.tile(x0i, (N + 1, 1)); ul = np.zeros((N, 4))
    qp = O.linearize(c, xl, ul, yr, ye_)
    dx, du = xo - xl, uo - ul
    J = 0.0
    for k in range(N):
        J += 0.5 * dx[k] @ (qp["Qd"][k] * dx[k]) + qp["q"][k] @ dx[k] + 0.5 * du[k] @ (qp["Rd"][k] * du[k]) + qp["r"][k] @ du[k]
    J += 0.5 * dx[N] @ (qp["Qd"][N] * dx[N]) + qp["q"][N] @ dx[N]
    res = max(np.abs(dx[k + 1] - qp["A"][k] @ dx[k] - qp["B"][k] @ du[k] - qp["b"][k]).max() for k in range(N))
    viol = max(0.0, (qp["lo"] - du).max(), (du - qp["hi"]).max())
    return J, res, viol, np.abs(qp["A"]).max(), np.abs(qp["B"]).max()


c.qp_polish = 1
both = (out["status"] == 0) & (ref["status"] == 0)
dd = np.where(both, np.abs(out["u0"] - ref["u0"]).max(1), -1.0)
print("default path, both status 0: worst |du0|", dd.max())
for i in np.argsort(-dd)[:4]:
    yr_i = yref if yref.ndim == 2 else yref[i]; ye_i = ye if ye.ndim == 1 else ye[i]
    jg = qp_check(c, x0[i], yr_i, ye_i, out["x"][i], out["u"][i]); jo = qp_check(c, x0[i], yr_i, ye_i, ref["x"][i], ref["u"][i])
    print(f"inst {i} |du0| {dd[i]:.2e}: gpu J {jg[0]:.10e} dyn res {jg[1]:.1e} viol {jg[2]:.1e} | oracle J {jo[0]:.10e} dyn res {jo[1]:.1e} viol {jo[2]:.1e} | max|A| {jg[3]:.1e} max|B| {jg[4]:.1e}")

# warm-started second solve, each side from its own first solution (as fuzz_parity.py does)
c.qp_polish = 1
out2 = s.solve_batch(x0, yref, ye, x_init=out["x"], u_init=out["u"], want_traj=True)
ref2 = O.solve_batch(c, x0, yref, ye, x_init=ref["x"], u_init=ref["u"], want_traj=True, nthreads=16)
mm = np.nonzero(out2["status"] != ref2["status"])[0]
print("warm solve: gpu", np.bincount(out2["status"], minlength=5), "oracle", np.bincount(ref2["status"], minlength=5), "mismatching instances", mm[:10])
for i in mm[:4]:
    print(f"  inst {i}: cold status gpu {out['status'][i]} oracle {ref['status'][i]} | warm gpu {out2['status'][i]} oracle {ref2['status'][i]} iters(oracle) {ref2['iters'][i]} "
          f"max|x_init| gpu {np.abs(out['x'][i]).max():.3g} oracle {np.abs(ref['x'][i]).max():.3g} |dx_init| {np.abs(out['x'][i] - ref['x'][i]).max():.2e}")
