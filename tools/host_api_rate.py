#!/usr/bin/env python3
"""Rate of the host-buffer entry point nmpc_solve_batch (PCIe inclusive: inputs and results cross the bus
on every call, pageable host memory, synchronous copies) next to the device-resident rate of bench.py."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402,F401

from rotors_mpc_controller_amd import _lib  # noqa: E402
from rotors_mpc_controller_amd.solver import NmpcOcpSolver  # noqa: E402
from rotors_mpc_controller_amd.synthetic import NEAR_HOVER, hover_reference, sample_x0  # noqa: E402
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)

for B in (4096, 65536):
    s = NmpcOcpSolver(_lib.default_config(max_batch=B))
    s.set_timing(False)
    x0 = sample_x0(B, 0, **NEAR_HOVER)
    yref, ye = hover_reference(s.config.N, s.config.mass * s.config.gravity / 4.0)
    yb, yeb = np.tile(yref, (B, 1, 1)), np.tile(ye, (B, 1))
    for name, (yr_, ye_) in (("per-instance yref", (yb, yeb)), ("broadcast yref", (yref, ye))):
        s.solve_batch(x0, yr_, ye_)
        n = 20
        t = time.perf_counter()
        for _ in range(n):
            s.solve_batch(x0, yr_, ye_)
        dt = (time.perf_counter() - t) / n
        mb = (x0.nbytes + yr_.nbytes + ye_.nbytes + B * 36) / 1e6
        print(f"nmpc_solve_batch B={B} {name}: {dt * 1e3:.3f} ms per call, {B / dt / 1e6:.2f} M solves/s, {mb / dt / 1e3:.1f} GB/s over the bus")
