"""How the active-set passes of the CPU oracle move the pin set of a long-horizon instance (round 5, config 5: N = 600, near-hover x0).
CPU only (oracle + ORC_POLISH_TRACE): the kernels make the same passes (pass counts are held equal in the GPU suite).
    python tools/dev/pin_trace.py [--horizon 600] [--batch 256] [--show 3]
"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from oracle import oracle as O  # noqa: E402
from rotors_mpc_controller_amd.synthetic import NEAR_HOVER, hover_reference, sample_x0  # noqa: E402
from tests.fuzz_draws import oracle_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--horizon", type=int, default=600)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--show", type=int, default=3, help="trace this many of the instances with the most passes")
    a = ap.parse_args()
    c = oracle_config(dict(N=a.horizon))
    x0 = sample_x0(1024, 0, **NEAR_HOVER)[:a.batch]
    yr, ye = hover_reference(a.horizon, c.mass * c.gravity / 4.0)
    os.environ.pop("ORC_POLISH_TRACE", None)
    out = O.solve_batch(c, x0, yr, ye, nthreads=8)
    print(f"N = {a.horizon}, first {a.batch} instances of the config-5 sample: passes histogram {np.bincount(out['passes']).tolist()}, "
          f"status != 0: {int((out['status'] != 0).sum())}", flush=True)
    os.environ["ORC_POLISH_TRACE"] = "1"
    for i in np.argsort(-out["passes"], kind="stable")[:a.show]:
        print(f"instance {i}: {out['passes'][i]} passes", flush=True)
        O.solve_batch(c, x0[i:i + 1], yr, ye, nthreads=1)
        sys.stderr.flush()


if __name__ == "__main__":
    main()
