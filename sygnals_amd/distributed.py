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


class RootGather:
    """Pipelined gather of equally shaped per-rank result blocks to `dst`, for loops that produce one block per
    step: `start(block)` issues the collective asynchronously (RCCL runs it on its own stream, so it overlaps the
    next step's kernels), `finish()` waits for it and returns the assembled [n_total, ...] tensor on the root
    (None elsewhere).  Receive buffers are allocated once and double-buffered; at most one gather is in flight.
    """

    def __init__(self, n_total: int, block_shape, dtype, device, dst: int = 0):
        self.active = dist.is_initialized() and dist.get_world_size() > 1
        self.dst = dst
        self.n_total = n_total
        self.work = None
        self.slot = 0
        self._held = None
        if not self.active:
            return
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.sizes = [shard_range(n_total, r, self.world) for r in range(self.world)]
        self.bmax = max(hi - lo for lo, hi in self.sizes)
        self.even = all(hi - lo == self.bmax for lo, hi in self.sizes)
        tail = tuple(block_shape[1:])
        if self.rank == dst:
            self.recv = [torch.empty((self.world, self.bmax) + tail, dtype=dtype, device=device) for _ in range(2)]
        self.pad = None if self.even else [torch.zeros((self.bmax,) + tail, dtype=dtype, device=device) for _ in range(2)]

    def start(self, local: torch.Tensor):
        """Begin gathering `local` (this rank's block).  A previous gather must have been finished."""
        if not self.active:
            self._held = local
            return
        assert self.work is None, "RootGather.start: the previous gather was not finished"
        lo, hi = self.sizes[self.rank]
        if local.shape[0] != hi - lo:
            raise ValueError(f"rank {self.rank} owns {hi - lo} items, got a block of {local.shape[0]}")
        send = local.contiguous()
        if not self.even:
            send = self.pad[self.slot]
            send[: local.shape[0]] = local
        bufs = list(self.recv[self.slot].unbind(0)) if self.rank == self.dst else None
        self._held = send                      # keep the send buffer alive until the collective has run
        self.work = dist.gather(send, bufs, dst=self.dst, async_op=True)

    def finish(self) -> Optional[torch.Tensor]:
        if not self.active:
            out, self._held = self._held, None
            return out
        if self.work is None:
            return None
        self.work.wait()
        self.work = None
        slot, self.slot = self.slot, self.slot ^ 1
        self._held = None
        if self.rank != self.dst:
            return None
        buf = self.recv[slot]
        if self.even:
            return buf.reshape((self.world * self.bmax,) + tuple(buf.shape[2:]))
        return torch.cat([buf[r, : hi - lo] for r, (lo, hi) in enumerate(self.sizes)], dim=0)


def run_sharded(clips_of_rank: Callable[[int, int], torch.Tensor], compute: Callable[[torch.Tensor], torch.Tensor],
                n_total: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Shard `n_total` clips over the ranks: each rank materialises its own block through
    `clips_of_rank(lo, hi)`, runs `compute` on it and the results are gathered on `dst`."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(n_total, rank, world)
    out = compute(clips_of_rank(lo, hi))
    return gather_to_root(out, n_total, dst)
