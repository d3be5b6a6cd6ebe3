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
        self.x = z(B, 13)
        self.u0, self.status = z(B, 4), torch.zeros(B, dtype=torch.int32, device=dev)
        self.xt = [z(B, N + 1, 13), z(B, N + 1, 13)]      # ping-pong warm-start trajectories
        self.ut = [z(B, N, 4), z(B, N, 4)]
        self.yref, self.yref_e = z(B, N, 17), z(B, 13)
        self.pos, self.yaw = z(B, 3), z(B)
        self.hover = cfg.mass * cfg.gravity / 4.0
        self.held = z(B, 4)       # the command the plant flies (the node's _last_command); starts at hover thrust

    def _tick(self, t: int, stream: int, warm: bool) -> None:
        """One control tick on `stream`: solve (warm-started from the other trajectory buffer), plant step."""
        s, B = self.s, self.B
        cur, prev = t & 1, (t + 1) & 1
        s.solve_batch_device(B, self.x.data_ptr(), self.yref.data_ptr(), self.yref_e.data_ptr(), False,
                             self.u0.data_ptr(), status_ptr=self.status.data_ptr(),
                             x_init_ptr=self.xt[prev].data_ptr() if warm else 0,
                             u_init_ptr=self.ut[prev].data_ptr() if warm else 0,
                             x_out_ptr=self.xt[cur].data_ptr(), u_out_ptr=self.ut[cur].data_ptr(), stream=stream)
        # a failed solve returns zeros (controller.py:448-450) and the node keeps flying its last command
        # (nodes/mpc_controller_node:124-129); the solver has handed back the cold-start point for such an
        # instance, so its next tick restarts cold although the launch is a warm-started one
        # (one launch; the state is advanced in place - fixed buffers, so the tick can be replayed from a HIP graph)
        s.hold_and_step_device(B, self.u0.data_ptr(), self.status.data_ptr(), self.held.data_ptr(), self.x.data_ptr(),
                               True, stream)

    def run(self, x0: np.ndarray, steps: int, setpoint=(0.0, 0.0, 1.0), yaw: float = 0.0, log: bool = True,
            use_graph: bool = False):
        """Returns host arrays xs [steps+1,B,13], us [steps,B,4] (log=False: only the final state and input).

        use_graph: after the cold first tick the loop is launch-bound (a tick is a fraction of a millisecond
        of kernels), so two ticks - one per warm-start buffer parity - are captured into a HIP graph and
        replayed; the library's per-solve timing events must be off (`solver.set_timing(False)`)."""
        torch, s, B = self.torch, self.s, self.B
        self.x.copy_(torch.as_tensor(np.ascontiguousarray(x0), dtype=self.dt_t))
        self.pos.copy_(torch.as_tensor(np.tile(np.asarray(setpoint, float), (B, 1)), dtype=self.dt_t))
        self.yaw.fill_(float(yaw))
        self.held.fill_(self.hover)
        side = torch.cuda.Stream(self.x.device) if use_graph else torch.cuda.current_stream()
        side.wait_stream(torch.cuda.current_stream())
        xs = torch.empty(steps + 1 if log else 1, B, 13, dtype=self.dt_t, device=self.x.device)
        us = torch.empty(steps if log else 1, B, 4, dtype=self.dt_t, device=self.x.device)
        with torch.cuda.stream(side):
            stream = side.cuda_stream
            s.build_hover_reference_device(B, self.pos.data_ptr(), self.yaw.data_ptr(), self.hover,
                                           self.yref.data_ptr(), self.yref_e.data_ptr(), stream)
            if log:
                xs[0].copy_(self.x)
            graph = None
            t = 0
            while t < steps:
                if use_graph and t >= 1 and (t & 1) == 1 and t + 2 <= steps and not log:
                    if graph is None:
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph, stream=side):
                            self._tick(t, side.cuda_stream, True)
                            self._tick(t + 1, side.cuda_stream, True)
                    graph.replay()
                    t += 2
                    continue
                self._tick(t, stream, t > 0)
                if log:
                    us[t].copy_(self.u0)
                    xs[t + 1].copy_(self.x)
                t += 1
            if not log:
                xs[0].copy_(self.x)
                us[0].copy_(self.u0)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return xs.cpu().numpy().astype(np.float64), us.cpu().numpy().astype(np.float64)
