#!/usr/bin/env python3
"""Diagnostic: K back-to-back solves of one batch, eager launches against one HIP-graph replay of the same K solves."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.solver import NmpcOcpSolver
from rotors_mpc_controller_amd.synthetic import NEAR_HOVER, hover_reference, sample_x0
import tools.dev._banner  # noqa: F401,E402  (first line of output: which binary runs)

B, K = 4096, 50
cfg = _lib.default_config(max_batch=B)
s = NmpcOcpSolver(cfg); s.set_timing(False)
yref, ye = hover_reference(cfg.N, cfg.mass * cfg.gravity / 4)
dev = torch.device("cuda", 0)
x0 = torch.from_numpy(sample_x0(B, 0, **NEAR_HOVER)).to(dev)
yr = torch.from_numpy(np.broadcast_to(yref, (B,) + yref.shape).copy()).to(dev); ye_d = torch.from_numpy(np.broadcast_to(ye, (B, 13)).copy()).to(dev)
u0 = torch.zeros(B, 4, dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
side = torch.cuda.Stream(dev)
def run(stream):
    for _ in range(K):
        s.solve_batch_device(B, x0.data_ptr(), yr.data_ptr(), ye_d.data_ptr(), False, u0.data_ptr(), status_ptr=st.data_ptr(), stream=stream)
with torch.cuda.stream(side):
    run(side.cuda_stream); torch.cuda.synchronize()
    t = time.perf_counter(); run(side.cuda_stream); run(side.cuda_stream); torch.cuda.synchronize(); e = (time.perf_counter() - t) / (2 * K)
    print(f"eager: {e*1e3:.4f} ms per step, {B/e/1e6:.2f} M solves/s")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        run(side.cuda_stream)
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter(); g.replay(); g.replay(); torch.cuda.synchronize(); e = (time.perf_counter() - t) / (2 * K)
    print(f"graph: {e*1e3:.4f} ms per step, {B/e/1e6:.2f} M solves/s")
