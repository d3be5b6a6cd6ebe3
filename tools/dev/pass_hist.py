#!/usr/bin/env python3
"""Histogram of active-set passes per instance (long-horizon workloads).  usage: python tools/dev/pass_hist.py [N] [B] [seed]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from rotors_mpc_controller_amd import _lib  # noqa: E402
from rotors_mpc_controller_amd.solver import NmpcOcpSolver  # noqa: E402
from rotors_mpc_controller_amd.synthetic import NEAR_HOVER, hover_reference, sample_x0  # noqa: E402
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
s = NmpcOcpSolver(_lib.default_config(N=N, max_batch=B))
yref, ye = hover_reference(N, s.config.mass * s.config.gravity / 4.0)
out = s.solve_batch(sample_x0(B, seed, **NEAR_HOVER), yref, ye)
it, ps = s.counts()
print("status", np.bincount(out["status"], minlength=5), "iterations", np.bincount(it))
h = np.bincount(np.abs(ps))
print("passes histogram:", {i: int(h[i]) for i in range(len(h)) if h[i]})
print("still active after pass p:", {p: int((np.abs(ps) > p).sum()) for p in range(1, 16)})
print(s.tail_states(B)[0], np.bincount(s.tail_states(B)[1], minlength=6))
