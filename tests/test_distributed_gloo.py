"""N>1 path on CPU: world_size-2 gloo.  The per-rank compute is the test-only host build of the
kernel bodies (tests/hostsim); what is under test is the sharding and the all-gather of u0."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rotors_mpc_controller_amd import _lib
from rotors_mpc_controller_amd.distributed import shard_bounds, solve_sharded
from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, hover_reference, sample_x0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import hostsim as H
    cfg = _lib.default_config()
    yref, ye = hover_reference(cfg.N, cfg.mass * cfg.gravity / 4.0)
    x0 = sample_x0(total, 11, **AGGRESSIVE)

    def solve(x):
        out = H.solve_batch(cfg, x, yref, ye)
        return out["u0"], out["status"]
    u0, st = solve_sharded(solve, x0)
    if rank == 0:
        q.put((u0, st))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_the_batch():
    for total, world in ((4096, 8), (37, 2), (5, 8), (64, 3)):
        spans = [shard_bounds(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_world_size_2_gather_equals_single_process():
    total = 37                                   # ragged: 19 + 18
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    u0, st = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from tests import hostsim as H
    cfg = _lib.default_config()
    yref, ye = hover_reference(cfg.N, cfg.mass * cfg.gravity / 4.0)
    ref = H.solve_batch(cfg, sample_x0(total, 11, **AGGRESSIVE), yref, ye)
    np.testing.assert_array_equal(st, ref["status"])
    np.testing.assert_array_equal(u0, ref["u0"])
