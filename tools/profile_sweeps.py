#!/usr/bin/env python3
"""Diagnostic: where does k_ipm spend its wall time?  Uses the NMPC_PROFILE build
(`make -C rotors_mpc_controller_amd/csrc prof`), whose kernel accumulates s_memrealtime
stamps per sweep.  Read the SHARES, not the absolute length (stamps perturb the schedule).

usage: python tools/profile_sweeps.py [--batch 4096] [--dtype f64] [--no-share]
"""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["ROTORS_NMPC_LIB"] = os.environ.get("NMPC_PROF_LIB", str(ROOT / "rotors_mpc_controller_amd" / "librotors_nmpc_hip_prof.so"))

import torch  # noqa: E402

from rotors_mpc_controller_amd import _lib  # noqa: E402
from rotors_mpc_controller_amd.solver import NmpcOcpSolver  # noqa: E402
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, hover_reference, sample_x0  # noqa: E402
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--no-share", action="store_true")
ap.add_argument("--mapping", default="team")
ap.add_argument("--dist", default="near_hover")
ap.add_argument("--no-polish", action="store_true")
ap.add_argument("--polish-passes", type=int, default=0)
a = ap.parse_args()

B = a.batch
cfg = _lib.default_config(max_batch=B, dtype=_lib.DTYPE_F64 if a.dtype == "f64" else _lib.DTYPE_F32,
                          flags=(0 if a.no_share else 1) | (2 if a.mapping == 'team' else 0))
if a.no_polish:
    cfg.update(qp_polish=0)
if a.polish_passes:
    cfg.update(qp_polish_passes=a.polish_passes, qp_polish_budget=a.polish_passes)
if a.dtype == "f32":
    cfg.update(qp_tol_comp=1e-8, qp_tol_stat=1e-6, qp_iter_max=30)
s = NmpcOcpSolver(cfg)
x0 = sample_x0(B, 0, **(NEAR_HOVER if a.dist == "near_hover" else AGGRESSIVE))
yref, ye = hover_reference(cfg.N, cfg.mass * cfg.gravity / 4)
for _ in range(3):
    out = s.solve_batch(x0, yref, ye)
st = s.stats()
lib = _lib.load()
lib.nmpc_debug_prof_copy.argtypes = [C.c_void_p, C.c_void_p]
lib.nmpc_debug_prof_copy.restype = C.c_int
Bp = (B + 63) // 64 * 64
host = np.zeros((8, Bp), dtype=np.int64)
assert lib.nmpc_debug_prof_copy(s._h, host.ctypes.data) == Bp
grp = 64 if a.mapping == 'lane' else 4
per_wave = host[:, :B].reshape(8, -1, grp)[:, :, 0] * 0.01   # s_memrealtime ticks at 100 MHz -> us
names = ["A factor(bwd)", "B fwd affine", "D bwd homog", "E fwd homog", "F mu sweep", "C check + exit", "final sweep", "prepare + start point"]
tot = per_wave.sum(0)
print(f"batch {B} dtype {a.dtype} share={not a.no_share}: kernel {st['ms_solve']:.3f} ms, prepare {st['ms_prepare']:.3f} ms, "
      f"iters mean {st['iter_mean']:.2f} max {st['iter_max']}")
print(f"per-wave total: mean {tot.mean():.1f} us, max {tot.max():.1f} us")
for i, n in enumerate(names):
    print(f"  {n:16s} mean {per_wave[i].mean():9.1f} us  ({100 * per_wave[i].sum() / tot.sum():5.1f} %)")
imax = int(tot.argmax())
print(f"slowest wave {imax}: " + ", ".join(f"{n.split()[0]} {per_wave[i, imax]:.1f}" for i, n in enumerate(names) if per_wave[i, imax] > 0))
srt = np.sort(tot)
print("per-wave total percentiles (us): " + ", ".join(f"p{q} {srt[min(len(srt) - 1, int(q / 100 * len(srt)))]:.1f}" for q in (10, 50, 90, 99, 100)))
