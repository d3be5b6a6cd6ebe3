"""bench.py on the GPU box: the contract line of the default single-GPU run and the RCCL path of the multi-GPU run
rehearsed with one rank (the 8-GPU run itself is the driver's)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(extra_env, *args):
    env = dict(os.environ, **extra_env)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=600,
                         cwd=str(ROOT), env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, lines[:3]               # the contract: ONE JSON line on stdout (no library banner in front of it)
    return json.loads(lines[-1])


def test_default_line_carries_the_contract_fields():
    d = _run({}, "--steps", "20", "--warmup", "3", "--no-secondary")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02            # device time of a step cannot exceed its wall time
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert d["status_histogram"][0] == 4096


def test_rccl_path_with_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(NMPC_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    d = _run(env, "--steps", "200", "--warmup", "20", "--no-cpu-baseline")
    assert d["n_gpus"] == 1 and "all-gather" in d["config"]["parallelism"]
    assert "in-place all-gather per tick" in d["config"]["exchange"]
    assert d["secondary"]["gather_every_8"]["value"] > 0         # the batched exchange ran too
    assert d["status_histogram"][0] == 4096
    # one collective per tick (solve straight into the gather buffer, solve + all-gather on one stream, replayed from a HIP
    # graph where RCCL can be captured) must not cost more than a tenth of the rate without any exchange
    assert d["value"] >= 0.9 * d["secondary"]["no_exchange"]["value"], (d["value"], d["secondary"]["no_exchange"]["value"], d["config"]["exchange"])


def test_device_time_of_the_rccl_path_covers_every_tick():
    """20 steps against an 8-tick graph leave 4 ticks that are run eagerly: they must lie inside the two HIP events that give
    device_ms_per_step (round 3 ran them after the closing event: device time, roofline.kernel_ms and frac of every N > 1 line of
    the driver's 20-step runs were 20 % off).  Device time per step then sits just under the wall time per step."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(NMPC_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    d = _run(env, "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-secondary")
    assert "in-place all-gather per tick" in d["config"]["exchange"]
    assert 0.88 * d["ms_per_step"] <= d["device_ms_per_step"] <= 1.02 * d["ms_per_step"], (d["device_ms_per_step"], d["ms_per_step"])
