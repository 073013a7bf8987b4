"""Clip-level data parallelism over the GPUs of one node.

The path has no cross-clip arithmetic, so there is nothing to all-reduce: one process per GPU,
rank 0 loads + packs the checkpoint, the packed arena (one flat byte buffer) is shipped with a
SINGLE broadcast (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests), the
conditioning with one more small broadcast, and every rank denoises its own slice of the batch.
The noise for the whole job is drawn once per rank from the same CPU generator seed, so the
result is bit-identical to a single-GPU run of the full batch.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

from . import packers


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `total` clips owned by `rank` (remainder to the low ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_arena(arena: Optional[packers.Arena], device, src: int = 0) -> packers.Arena:
    """One collective for the whole model: layout table (python object) + the flat byte buffer."""
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta[0] = (int(arena.buffer.numel()), arena.table)
    dist.broadcast_object_list(meta, src=src)
    total, table = meta[0]
    if rank != src:
        arena = packers.Arena(total, table, device)
    dist.broadcast(arena.buffer, src=src)
    return arena


def broadcast_tensors(tensors: Optional[Dict[str, torch.Tensor]], device, src: int = 0) -> Dict[str, torch.Tensor]:
    """Small dict of fp32 tensors (conditioning) packed into one buffer, one broadcast."""
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta[0] = [(k, tuple(v.shape)) for k, v in tensors.items()]
    dist.broadcast_object_list(meta, src=src)
    n = sum(int(torch.tensor(s).prod()) if len(s) else 1 for _, s in meta[0])
    flat = torch.empty(n, dtype=torch.float32, device=device)
    if rank == src:
        torch.cat([tensors[k].reshape(-1).to(device=device, dtype=torch.float32) for k, _ in meta[0]], out=flat)
    dist.broadcast(flat, src=src)
    out, off = {}, 0
    for k, s in meta[0]:
        m = 1
        for d in s:
            m *= d
        out[k] = flat[off:off + m].view(s).clone()
        off += m
    return out
