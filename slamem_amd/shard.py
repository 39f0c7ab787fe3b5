"""Multi-GPU plumbing of the MEM path: one process per GPU over torch.distributed (backend "nccl" = RCCL on
ROCm, "gloo" in the CPU tests).  The path shards by query record with NO data-path collective:

* the index is built once on rank 0 and replicated with ONE broadcast of its arena (a single contiguous
  HBM buffer, see include/slamem_hip.h) -- xGMI is point-to-point, so one large transfer per peer beats
  many small ones;
* every rank matches a contiguous range of the query records (balanced by total bases);
* results stay on the rank that produced them; only per-rank MEM counts (and, on request, the
  variable-length MEM arrays) are gathered.  Concatenating in rank order restores input order because
  the ranges are contiguous.

Everything here is device-agnostic tensor plumbing; no compute.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


BROADCAST_PIECE = 1 << 30


def world() -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(offsets: np.ndarray, world_size: int) -> np.ndarray:
    """Split records 0..num-1 into world_size contiguous ranges with (almost) equal total bases.

    offsets: uint64[num+1] record start offsets.  Returns int64[world_size+1] record boundaries."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    num = offsets.shape[0] - 1
    total = int(offsets[-1] - offsets[0])
    bounds = np.zeros(world_size + 1, dtype=np.int64)
    bounds[-1] = num
    for r in range(1, world_size):
        target = int(offsets[0]) + (total * r) // world_size
        # first record whose start is >= target
        bounds[r] = int(np.searchsorted(offsets[:-1], np.uint64(target), side="left"))
    bounds = np.maximum.accumulate(bounds)
    bounds[-1] = num
    return bounds


def broadcast_arena(arena: torch.Tensor | None, device, src: int = 0, force: bool = False) -> torch.Tensor:
    """Replicate the index arena (uint8 tensor) from rank `src` to every rank: size first, then the bytes.
    force: issue the collectives even in a one-rank group (rehearsal of the RCCL calls on a one-GPU box)."""
    rank, ws = world()
    if ws == 1 and not (force and dist.is_initialized()):
        assert arena is not None
        return arena
    size = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        size[0] = arena.numel()
    dist.broadcast(size, src)
    if rank != src:
        arena = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    # one logical broadcast, issued in pieces of at most 1 GiB: the arena is 3 GB at 100 Mbp and 98 GB at 3.1 Gbp, more
    # elements than some collective front ends count in 32 bits; xGMI moves a GiB per call at full link rate
    total = arena.numel()
    for off in range(0, total, BROADCAST_PIECE):
        dist.broadcast(arena[off:min(total, off + BROADCAST_PIECE)], src)
    return arena


def gather_counts(count: int, device, force: bool = False, buffers=None) -> torch.Tensor:
    """All ranks learn every rank's MEM count (int64[world]).  buffers = (mine int64[1], out int64[world]) on `device`: a caller
    that gathers at every step hands in its own pair and pays no allocation and no host-to-device copy per call."""
    rank, ws = world()
    if buffers is not None:
        mine, out = buffers
        mine.fill_(int(count))
    else:
        mine = torch.tensor([count], dtype=torch.int64, device=device)
        out = None
    if ws == 1 and not (force and dist.is_initialized()):
        return mine
    if out is None:
        out = torch.zeros(ws, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine) if device.type == "cuda" else dist.all_gather(list(out.split(1)), mine)
    return out


def gather_variable(rows: torch.Tensor, counts: torch.Tensor, dst: int = 0) -> torch.Tensor | None:
    """Gather per-rank (count_r, k) int32 row blocks on rank `dst`, concatenated in rank order."""
    rank, ws = world()
    if ws == 1:
        return rows
    k = rows.shape[1]
    if rank == dst:
        parts = []
        for r in range(ws):
            if r == dst:
                parts.append(rows)
            else:
                buf = torch.empty((int(counts[r].item()), k), dtype=rows.dtype, device=rows.device)
                if buf.numel():
                    dist.recv(buf, src=r)
                parts.append(buf)
        return torch.cat(parts, dim=0)
    if rows.numel():
        dist.send(rows.contiguous(), dst=dst)
    return None
