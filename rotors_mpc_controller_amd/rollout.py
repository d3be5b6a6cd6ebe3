"""Closed-loop Monte-Carlo rollouts, entirely on the device.

B independent vehicles: solve -> apply the first-stage command to the plant (the controller's own
model and ERK scheme) -> renormalise the quaternion (controller.py:406-409) -> next solve warm-started
with the previous solution, NOT shifted (controller.py:419-424, 455-461).  Exercises per-stage
distinct linearisations (no cold-start sharing after the first tick).  PyTorch only holds the
device buffers; every arithmetic step is a kernel of librotors_nmpc_hip.so.
"""
from __future__ import annotations

import numpy as np

from .solver import NmpcOcpSolver


class ClosedLoopRollout:
    def __init__(self, solver: NmpcOcpSolver, batch: int):
        import torch
        self.torch = torch
        self.s, self.B, self.N = solver, int(batch), solver.N
        cfg = solver.config
        self.dt_t = torch.float64 if cfg.dtype == 0 else torch.float32
        dev = torch.device("cuda", cfg.device)
        z = lambda *shape: torch.zeros(*shape, dtype=self.dt_t, device=dev)  # noqa: E731
        B, N = self.B, self.N
        self.x, self.xn = z(B, 13), z(B, 13)
        self.u0, self.status = z(B, 4), torch.zeros(B, dtype=torch.int32, device=dev)
        self.xt = [z(B, N + 1, 13), z(B, N + 1, 13)]      # ping-pong warm-start trajectories
        self.ut = [z(B, N, 4), z(B, N, 4)]
        self.yref, self.yref_e = z(B, N, 17), z(B, 13)
        self.pos, self.yaw = z(B, 3), z(B)
        self.hover = cfg.mass * cfg.gravity / 4.0

    def run(self, x0: np.ndarray, steps: int, setpoint=(0.0, 0.0, 1.0), yaw: float = 0.0):
        """Returns host arrays xs [steps+1,B,13], us [steps,B,4]."""
        torch, s, B = self.torch, self.s, self.B
        stream = torch.cuda.current_stream().cuda_stream
        self.x.copy_(torch.as_tensor(np.ascontiguousarray(x0), dtype=self.dt_t))
        self.pos.copy_(torch.as_tensor(np.tile(np.asarray(setpoint, float), (B, 1)), dtype=self.dt_t))
        self.yaw.fill_(float(yaw))
        s.build_hover_reference_device(B, self.pos.data_ptr(), self.yaw.data_ptr(), self.hover,
                                       self.yref.data_ptr(), self.yref_e.data_ptr(), stream)
        xs = torch.empty(steps + 1, B, 13, dtype=self.dt_t, device=self.x.device)
        us = torch.empty(steps, B, 4, dtype=self.dt_t, device=self.x.device)
        xs[0].copy_(self.x)
        for t in range(steps):
            cur, prev = t & 1, (t + 1) & 1
            warm = t > 0
            s.solve_batch_device(B, self.x.data_ptr(), self.yref.data_ptr(), self.yref_e.data_ptr(), False,
                                 self.u0.data_ptr(), status_ptr=self.status.data_ptr(),
                                 x_init_ptr=self.xt[prev].data_ptr() if warm else 0,
                                 u_init_ptr=self.ut[prev].data_ptr() if warm else 0,
                                 x_out_ptr=self.xt[cur].data_ptr(), u_out_ptr=self.ut[cur].data_ptr(), stream=stream)
            s.plant_step_device(B, self.x.data_ptr(), self.u0.data_ptr(), self.xn.data_ptr(), True, stream)
            us[t].copy_(self.u0)
            self.x, self.xn = self.xn, self.x
            xs[t + 1].copy_(self.x)
        torch.cuda.synchronize()
        return xs.cpu().numpy().astype(np.float64), us.cpu().numpy().astype(np.float64)
