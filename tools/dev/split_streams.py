"""Experiment (round 5, config 5): one batch of B instances solved as G groups of B / G on G handles and G streams at the same time.
The block-parallel tail of ONE handle runs its listed instances in lock step - every step costs max(its throughput time, the dependency
latency of one instance) - so the long early steps of one group could run beside the latency-bound late steps of another.
    python tools/dev/split_streams.py --batch 1024 --horizon 600 --groups 1 2 4
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import tools.dev._banner  # noqa: E402,F401  (prints nmpc_version() first)
import torch  # noqa: E402

from rotors_mpc_controller_amd import _lib  # noqa: E402
from rotors_mpc_controller_amd.pipeline import BatchPipeline  # noqa: E402
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, hover_reference, sample_x0  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--horizon", type=int, default=600)
    ap.add_argument("--groups", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--dist", default="near_hover")
    ap.add_argument("--offset-us", type=float, default=0.0, help="group g starts g * offset later (a spin kernel on its stream): out of phase, "
                    "the latency-bound launches of one group run beside the throughput-bound ones of another")
    ap.add_argument("--interleave", action="store_true", help="group g takes instances g, g + G, ... instead of a contiguous slice")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, N = a.batch, a.horizon
    base = _lib.default_config(N=N, max_batch=B, device=0, flags=_lib.FLAG_SHARE_COLD_START | _lib.FLAG_TEAM_MAPPING)
    hover = base.mass * base.gravity / 4.0
    x0_h = sample_x0(B, 0, **(NEAR_HOVER if a.dist == "near_hover" else AGGRESSIVE))
    yref_h, ye_h = hover_reference(N, hover)
    yref = torch.from_numpy(yref_h).to(dev)
    ye = torch.from_numpy(ye_h).to(dev)
    ref_u0 = None
    # spin kernel calibration: cycles per microsecond of torch.cuda._sleep
    torch.cuda._sleep(1000); torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize(dev)
    cyc_per_us = 10_000_000 / (1e3 * e0.elapsed_time(e1))
    print(f"spin kernel: {cyc_per_us:.1f} cycles per us", flush=True)
    for G in a.groups:
        assert B % G == 0
        Bg = B // G
        cfg = _lib.default_config(N=N, max_batch=Bg, device=0, flags=_lib.FLAG_SHARE_COLD_START | _lib.FLAG_TEAM_MAPPING)
        pipe = BatchPipeline(cfg, depth=G)
        idx = [np.arange(g, B, G) if a.interleave else np.arange(g * Bg, (g + 1) * Bg) for g in range(G)]
        x0 = [torch.from_numpy(np.ascontiguousarray(x0_h[i])).to(dev) for i in idx]
        u0 = [torch.zeros(Bg, 4, dtype=torch.float64, device=dev) for _ in range(G)]
        st = [torch.zeros(Bg, dtype=torch.int32, device=dev) for _ in range(G)]

        def step():
            for g in range(G):
                if a.offset_us > 0 and g > 0:
                    with torch.cuda.stream(pipe.streams[g]):
                        torch.cuda._sleep(int(g * a.offset_us * cyc_per_us))
                pipe.submit(Bg, x0[g].data_ptr(), yref.data_ptr(), ye.data_ptr(), True, u0[g].data_ptr(), status_ptr=st[g].data_ptr(),
                            after_current_stream=False)
            pipe.synchronize()              # a step = the whole batch done (the next batch of a closed loop needs it)
        for _ in range(2):
            step()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(a.steps):
            step()
        el = (time.perf_counter() - t) / a.steps
        full = np.zeros((B, 4))
        for g in range(G):
            full[idx[g]] = u0[g].cpu().numpy()
        bad = sum(int(s.abs().sum()) for s in st)
        if ref_u0 is None:
            ref_u0 = full
        print(f"groups {G} offset {a.offset_us:.0f} us: {1e3 * el:7.3f} ms per batch of {B} (N = {N})   {B / el / 1e3:8.1f} k solves/s   status != 0: {bad}   "
              f"max |u0 - one group| {np.abs(full - ref_u0).max():.1e}", flush=True)
        pipe.close()


if __name__ == "__main__":
    main()
