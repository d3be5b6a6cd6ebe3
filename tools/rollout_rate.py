#!/usr/bin/env python3
"""Closed-loop Monte-Carlo rollout rate: B vehicles, T control ticks, everything on the
device (solve warm-started from the previous solution -> plant step -> next tick).
usage: python tools/rollout_rate.py [--batch 4096] [--ticks 200] [--dist aggressive]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402,F401

from rotors_mpc_controller_amd import _lib  # noqa: E402
from rotors_mpc_controller_amd.rollout import ClosedLoopRollout  # noqa: E402
from rotors_mpc_controller_amd.solver import NmpcOcpSolver  # noqa: E402
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, sample_x0  # noqa: E402
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--ticks", type=int, default=200)
ap.add_argument("--dist", default="aggressive")
ap.add_argument("--graph", action="store_true", help="replay the tick pair from a HIP graph (no per-tick logging)")
a = ap.parse_args()
s = NmpcOcpSolver(_lib.default_config(max_batch=a.batch))
s.set_timing(False)
ro = ClosedLoopRollout(s, a.batch)
x0 = sample_x0(a.batch, 5, **(AGGRESSIVE if a.dist == "aggressive" else NEAR_HOVER))
ro.run(x0, 5)                                   # warm-up
t = time.perf_counter()
xs, us = ro.run(x0, a.ticks, log=not a.graph, use_graph=a.graph)
dt = time.perf_counter() - t
err = np.linalg.norm(xs[-1][:, 0:3] - np.array([0.0, 0.0, 1.0]), axis=1)
print(f"closed loop: B={a.batch} ticks={a.ticks} ({a.dist} start): {dt * 1e3 / a.ticks:.3f} ms per tick "
      f"({'HIP graph replay, final state only' if a.graph else 'eager launches, per-tick logs copied at the end'}), {a.batch * a.ticks / dt / 1e6:.2f} M solves/s; "
      f"final position error median {np.median(err):.3e} m, max {err.max():.3e} m")
