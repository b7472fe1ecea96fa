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


def usable_cpus() -> int:
    """CPUs this process may really use: min(affinity mask, cgroup v2 cpu.max quota)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _parse_cpulist(text: str) -> list:
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def plan_rank_cpus(allowed: list, local_rank: int, local_world: int, numa_cpus: Optional[list] = None, numa_peers: int = 1,
                   numa_slot: int = 0) -> list:
    """CPU set for one rank of `local_world` ranks on a node (pure function: tests/test_host_logic.py).

    With the GPU's NUMA node known: the node's allowed CPUs, split evenly among the `numa_peers` ranks whose GPUs sit on the
    same node (this rank being number `numa_slot` of them).  Otherwise: an even contiguous split of the allowed CPUs over the
    local ranks.  Never returns an empty set: with fewer CPUs than ranks the whole allowed set is kept (no binding)."""
    allowed = sorted(allowed)
    if numa_cpus and 0 <= numa_slot < numa_peers:      # an inconsistent slot falls through to the even split (never an empty set)
        pool = [c for c in sorted(numa_cpus) if c in set(allowed)]
        if len(pool) >= numa_peers >= 1:
            per = len(pool) // numa_peers
            return pool[numa_slot * per:(numa_slot + 1) * per]
    if len(allowed) < local_world:
        return allowed
    per = len(allowed) // local_world
    return allowed[local_rank * per:(local_rank + 1) * per]


def gpu_numa_node(pci_bus_id, pci_domain_id: int = 0) -> Optional[int]:
    """NUMA node of the GPU at (domain, bus) from /sys, None when the topology is not exposed."""
    if pci_bus_id is None:
        return None
    try:
        node = int(open(f"/sys/bus/pci/devices/{int(pci_domain_id):04x}:{int(pci_bus_id):02x}:00.0/numa_node").read())
        return node if node >= 0 else None
    except (OSError, ValueError):
        return None


def bind_rank_to_cpus(local_rank: int, local_world: int, pci_bus_id=None, pci_domain_id: int = 0, gpu_numa_nodes: Optional[list] = None) -> dict:
    """Pin this process (and the library's copy threads it will spawn) to its share of the host cores: the cores of the GPU's
    NUMA node when /sys exposes it, an even split otherwise.  gpu_numa_nodes: NUMA node of every local rank's GPU (bench.py reads
    them for all local devices with gpu_numa_node — no collective needed): the ranks whose GPUs share a node split it in rank
    order, whatever the GPU-to-node mapping looks like (interleaved mappings included).  Without it each rank takes its node's
    cores divided by ceil(local_world / number of nodes), which is collision-free only when the ranks of a node are consecutive.
    A list that disagrees with this rank's own /sys reading is ignored (even split).  Returns what was done (for the bench JSON)."""
    import os
    allowed = sorted(os.sched_getaffinity(0))
    node, numa_cpus = gpu_numa_node(pci_bus_id, pci_domain_id), None
    if node is not None:
        try:
            numa_cpus = _parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
        except (OSError, ValueError):
            node, numa_cpus = None, None
    peers, slot = 1, 0
    if numa_cpus:
        if gpu_numa_nodes:
            same = [r for r, nd in enumerate(gpu_numa_nodes) if nd == node]
            if local_rank in same:
                peers, slot = len(same), same.index(local_rank)
            else:                                   # the list contradicts this rank's own reading: do not trust either
                numa_cpus = None
        else:
            try:
                n_nodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
            except OSError:
                n_nodes = 1
            peers = max(1, -(-local_world // max(n_nodes, 1)))
            slot = local_rank % peers
    cpus = plan_rank_cpus(allowed, local_rank, local_world, numa_cpus, peers, slot)
    bound = False
    if cpus and len(cpus) < len(allowed):
        os.sched_setaffinity(0, cpus)
        bound = True
    return {"bound": bound, "cpus_bound": len(cpus), "first_cpu": cpus[0] if cpus else None, "numa_node": node,
            "numa_peers": peers if numa_cpus else None, "numa_slot": slot if numa_cpus else None,
            "policy": "GPU's NUMA node" if numa_cpus else "even split of the allowed CPUs"}
