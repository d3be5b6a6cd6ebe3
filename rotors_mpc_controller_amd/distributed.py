"""Multi-GPU: independent Monte-Carlo instances sharded over ranks, one exchange of the commands.

The path shards trivially: instances share nothing but read-only constants, so each
rank solves a contiguous slice of the batch on its own GPU with no data-path collective, and the
first-stage commands ``u0 [B_loc, 4]`` (+ status) are all-gathered once per batch -- RCCL over
xGMI when the process group is "nccl", gloo on CPU in the tests.  128 KiB per rank at B_loc = 4096
in FP64: latency-bound, a single all-gather, no bucketing.  There is no counterpart in the
reference (single process, one instance; controller.py:56,162).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split; the first `total % world` ranks own one more instance."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_seed(rank: int) -> int:
    """Seed of the per-rank x0 shard in the weak-scaling configuration."""
    return 100 + rank


def all_gather_commands(u0: torch.Tensor, status: torch.Tensor, group=None):
    """u0 [B_loc,4], status [B_loc] (same B_loc on every rank) -> ([W*B_loc,4], [W*B_loc])."""
    world = dist.get_world_size(group)
    g_u = torch.empty((world * u0.shape[0], u0.shape[1]), dtype=u0.dtype, device=u0.device)
    g_s = torch.empty((world * status.shape[0],), dtype=status.dtype, device=status.device)
    dist.all_gather_into_tensor(g_u, u0.contiguous(), group=group)
    dist.all_gather_into_tensor(g_s, status.contiguous(), group=group)
    return g_u, g_s


def all_gather_ragged(u0: torch.Tensor, status: torch.Tensor, total: int, group=None):
    """Same for uneven shards (total not divisible by the world size): pad to the largest shard."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
    m = max(sizes)
    pu = torch.zeros((m, u0.shape[1]), dtype=u0.dtype, device=u0.device)
    ps = torch.full((m,), -1, dtype=status.dtype, device=status.device)
    pu[:sizes[rank]] = u0
    ps[:sizes[rank]] = status
    g_u, g_s = all_gather_commands(pu, ps, group)
    keep = torch.cat([torch.arange(r * m, r * m + sizes[r]) for r in range(world)]).to(u0.device)
    return g_u[keep], g_s[keep]


def solve_sharded(solve_fn: Callable[[np.ndarray], Tuple[np.ndarray, np.ndarray]], x0_full: np.ndarray,
                  device: Optional[torch.device] = None, group=None):
    """Each rank solves its contiguous slice of `x0_full` with `solve_fn(x0) -> (u0, status)` and
    all ranks end up with the commands of the whole batch, in the original order."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(x0_full.shape[0], world, rank)
    u0, status = solve_fn(x0_full[lo:hi])
    dev = device or torch.device("cpu")
    tu = torch.as_tensor(np.ascontiguousarray(u0), device=dev)
    ts = torch.as_tensor(np.ascontiguousarray(status), device=dev)
    g_u, g_s = all_gather_ragged(tu, ts, x0_full.shape[0], group)
    return g_u.cpu().numpy(), g_s.cpu().numpy()
