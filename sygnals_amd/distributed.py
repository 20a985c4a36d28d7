"""Clip-level sharding across the GPUs of one node (one process per GPU).

Clips are independent units: rank r of W owns the contiguous block
[r*B/W, (r+1)*B/W) and computes it with no data-path collective; the only exchange is the
final gather of the per-rank feature blocks to rank 0 (RCCL over xGMI when the process group
uses the "nccl" backend; "gloo" works for CPU rehearsal).  Each peer sends its block over its
own direct link to the root, which is the pattern torch.distributed.gather issues on RCCL.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced block of rank `rank` (sizes differ by at most one)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_to_root(local: torch.Tensor, n_total: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Gather per-rank blocks [b_r, ...] (contiguous shards of a leading axis of size n_total) to `dst`.

    Returns the assembled [n_total, ...] tensor on `dst`, None elsewhere.  Works with uneven shards
    (blocks are padded to the largest shard for the collective and trimmed on the root).
    """
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    bmax = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < bmax:
        pad = torch.zeros((bmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    pad = pad.contiguous()
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def run_sharded(clips_of_rank: Callable[[int, int], torch.Tensor], compute: Callable[[torch.Tensor], torch.Tensor],
                n_total: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Shard `n_total` clips over the ranks: each rank materialises its own block through
    `clips_of_rank(lo, hi)`, runs `compute` on it and the results are gathered on `dst`."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(n_total, rank, world)
    out = compute(clips_of_rank(lo, hi))
    return gather_to_root(out, n_total, dst)
