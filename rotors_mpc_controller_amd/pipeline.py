"""Independent batches in flight (Monte-Carlo use; no counterpart in the reference).

One solver handle serialises its solves: the kernels of a batch run one wave per SIMD and the batch ends with its
slowest wave - the few instances that need more active-set passes - while most of the chip idles.
Batches that do not depend on each other need not wait for that: `depth` handles, each with its own workspace and its
own HIP stream, take the batches in turn, and the waves of the next batch fill the SIMDs the finished waves leave
(B = 4096, FP64: 63 M -> 89 M solves/s with depth 2).  A closed loop cannot use this - its next solve needs this one's
result; the C ABI stays stream-ordered and one handle stays one solve at a time (include/rotors_nmpc.h).

PyTorch only provides the streams; every solve is nmpc_solve_batch_device of librotors_nmpc_hip.so.
"""
from __future__ import annotations

from .solver import NmpcOcpSolver


class BatchPipeline:
    def __init__(self, config, depth: int = 2):
        import torch
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self._torch = torch
        self._dev = torch.device("cuda", config.device)
        self.solvers = [NmpcOcpSolver(config) for _ in range(depth)]
        for s in self.solvers:
            s.set_timing(False)            # per-solve HIP events would serialise nothing but cost host time
        self.streams = [torch.cuda.Stream(self._dev) for _ in range(depth)]
        self._next = 0

    @property
    def depth(self) -> int:
        return len(self.solvers)

    def submit(self, B: int, x0_ptr: int, yref_ptr: int, yref_e_ptr: int, yref_bcast: bool, u0_ptr: int,
               status_ptr: int = 0, after_current_stream: bool = True, **kw) -> int:
        """Enqueue one batch on the next slot (round robin) and return the slot.  Arguments as
        NmpcOcpSolver.solve_batch_device.  The slot's stream first waits for the caller's current stream, so inputs
        produced there are complete (after_current_stream=False skips that event pair - ~8 us of host time per batch -
        when the inputs are known to be ready); the outputs are ready when `wait(slot)` / `synchronize()` returns.
        The buffers of a batch must stay untouched until then, and batches in flight must not share output buffers."""
        slot = self._next
        self._next = (slot + 1) % len(self.solvers)
        st = self.streams[slot]
        if after_current_stream:
            st.wait_stream(self._torch.cuda.current_stream(self._dev))
        self.solvers[slot].solve_batch_device(B, x0_ptr, yref_ptr, yref_e_ptr, yref_bcast, u0_ptr, status_ptr=status_ptr,
                                              stream=st.cuda_stream, **kw)
        return slot

    def wait(self, slot: int) -> None:
        """Make the caller's current stream wait for everything submitted to `slot` (no host synchronisation)."""
        self._torch.cuda.current_stream(self._dev).wait_stream(self.streams[slot])

    def synchronize(self) -> None:
        for st in self.streams:
            st.synchronize()

    def close(self) -> None:
        self.synchronize()
        for s in self.solvers:
            s.close()
        self.solvers = []
