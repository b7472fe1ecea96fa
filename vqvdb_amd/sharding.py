"""Leaf sharding across ranks (one process per GPU).

The path is embarrassingly parallel over leaves (no op mixes information across leaves:
GroupNorm and channel attention are per-sample, python/VQVAE_v2.py:190-228), so multi-GPU is
pure partitioning: rank g takes the contiguous leaf range
    [g*ceil(B/G), min(B, (g+1)*ceil(B/G)))            
weights/codebook (4.1 MB) are replicated, and there is NO data-path collective.  The only
communication is outside the hot path: the bench's barrier/max-over-ranks timing and an
optional gather of per-rank results onto rank 0 for callers that want one buffer.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def shard_range(n_leaves: int, rank: int, world: int) -> Tuple[int, int]:
    per = -(-n_leaves // world)
    lo = min(n_leaves, rank * per)
    return lo, min(n_leaves, lo + per)


def max_over_ranks(seconds: float, device=None) -> float:
    """Wall time of the slowest rank (the bench's timing rule)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def encode_shard(codec, leaves: np.ndarray, rank: int, world: int) -> np.ndarray:
    """Encode this rank's leaf range of a batch every rank can see; returns [hi-lo, 64] uint8."""
    lo, hi = shard_range(len(leaves), rank, world)
    if hi == lo:
        return np.empty((0, 64), dtype=np.uint8)
    return codec.encode(leaves[lo:hi])


def decode_shard(codec, indices: np.ndarray, rank: int, world: int) -> np.ndarray:
    lo, hi = shard_range(len(indices), rank, world)
    if hi == lo:
        return np.empty((0, 512), dtype=np.float32)
    return codec.decode(indices[lo:hi])


def gather_to_rank0(local: np.ndarray, n_total: int, rank: int, world: int) -> Optional[np.ndarray]:
    """Assemble per-rank shard results in leaf order on rank 0 (outside the timed hot path)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local
    per = -(-n_total // world)
    pad = np.zeros((per,) + local.shape[1:], dtype=local.dtype)
    pad[:len(local)] = local
    t = torch.from_numpy(pad)
    bufs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, bufs, dst=0)
    if rank != 0:
        return None
    parts = []
    for g in range(world):
        lo, hi = shard_range(n_total, g, world)
        parts.append(bufs[g].numpy()[:hi - lo])
    return np.concatenate(parts)
