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


def _worker_cfg4(rank, world, port, q):
    """BASELINE config 4 in miniature on CPU: equal shards of 4096 (seed 100 + rank), no data-path collective, one
    all-gather of u0 / status per tick (all_gather_commands) and the grouped form bench.py uses (G ticks per collective,
    asynchronous handle, [world][G][B][4] layout)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from rotors_mpc_controller_amd.distributed import all_gather_commands, rank_seed
    from rotors_mpc_controller_amd.synthetic import NEAR_HOVER
    B = 4096
    c = O.default_config(qp_gamma=0.0, qp_polish=1)
    yref, ye = O.hover_yref(c)
    out = O.solve_batch(c, sample_x0(B, rank_seed(rank), **NEAR_HOVER), yref, ye, nthreads=3)
    u0, st = torch.from_numpy(out["u0"]), torch.from_numpy(out["status"].astype(np.int32))
    g_u, g_s = all_gather_commands(u0, st)
    # grouped exchange: G consecutive ticks share one asynchronous all-gather (tick t carries u0 + t here)
    G = 3
    grp = torch.stack([u0 + float(t) for t in range(G)])                       # [G][B][4]
    gathered = torch.zeros(world, G, B, 4, dtype=torch.float64)
    h = dist.all_gather_into_tensor(gathered.view(world * G * B, 4), grp.view(G * B, 4), async_op=True)
    h.wait()
    if rank == 0:
        q.put((g_u.numpy(), g_s.numpy(), gathered.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_equal_shards_of_4096_all_gather_commands_and_grouped_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_cfg4, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g_u, g_s, grouped = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import oracle as O
    from rotors_mpc_controller_amd.distributed import rank_seed
    from rotors_mpc_controller_amd.synthetic import NEAR_HOVER
    c = O.default_config(qp_gamma=0.0, qp_polish=1)
    yref, ye = O.hover_yref(c)
    assert g_u.shape == (2 * 4096, 4) and g_s.shape == (2 * 4096,)
    for r in range(2):
        ref = O.solve_batch(c, sample_x0(4096, rank_seed(r), **NEAR_HOVER), yref, ye, nthreads=6)
        np.testing.assert_array_equal(g_u[r * 4096:(r + 1) * 4096], ref["u0"])       # rank r's block, original order
        np.testing.assert_array_equal(g_s[r * 4096:(r + 1) * 4096], ref["status"])
        for t in range(3):
            np.testing.assert_array_equal(grouped[r, t], ref["u0"] + float(t))       # [world][G][B][4], as bench.py indexes it


def _worker_inplace(rank, world, port, q):
    """The default exchange of bench.py's multi-rank run, on gloo: every rank's solve writes its commands straight into ITS slot of
    a gather buffer [world][B][4], the all-gather is in place (input = that slot of the output), two buffers alternate between
    ticks.  Same tensor expressions as bench.py (gat2[b][rank], gat2[b].view(world * B, 4))."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import hostsim as H
    from rotors_mpc_controller_amd.distributed import rank_seed
    from rotors_mpc_controller_amd.synthetic import NEAR_HOVER
    B = 96
    cfg = _lib.default_config()
    yref, ye = hover_reference(cfg.N, cfg.mass * cfg.gravity / 4.0)
    u0 = torch.from_numpy(H.solve_batch(cfg, sample_x0(B, rank_seed(rank), **NEAR_HOVER), yref, ye)["u0"])
    gat2 = [torch.full((world, B, 4), float("nan"), dtype=torch.float64) for _ in range(2)]
    seen = []
    for t in range(4):                           # four ticks: buffers 0, 1, 0, 1; tick t carries u0 + t
        b = t & 1
        gat2[b][rank].copy_(u0 + float(t))       # "the solve writes into this rank's slot"
        dist.all_gather_into_tensor(gat2[b].view(world * B, 4), gat2[b][rank])
        seen.append(gat2[b].clone())
    if rank == 0:
        q.put(torch.stack(seen).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_in_place_gather_buffer_layout_of_the_default_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_inplace, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    seen = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from tests import hostsim as H
    from rotors_mpc_controller_amd.distributed import rank_seed
    from rotors_mpc_controller_amd.synthetic import NEAR_HOVER
    cfg = _lib.default_config()
    yref, ye = hover_reference(cfg.N, cfg.mass * cfg.gravity / 4.0)
    assert seen.shape == (4, 2, 96, 4)
    for r in range(2):
        ref = H.solve_batch(cfg, sample_x0(96, rank_seed(r), **NEAR_HOVER), yref, ye)["u0"]
        for t in range(4):
            np.testing.assert_array_equal(seen[t, r], ref + float(t))      # slot r of tick t's buffer = rank r's commands of that tick
