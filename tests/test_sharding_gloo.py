"""Multi-rank host logic on CPU: world_size-2 gloo.  The HIP codec cannot run here, so each rank
gets a FAKE backend (the CPU oracle behind the same encode/decode signature); what is tested is
the product's sharding/gather/timing code in vqvdb_amd/sharding.py."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_exactly():
    from vqvdb_amd.sharding import shard_range
    for n in (0, 1, 7, 64, 65, 65536, 1000003):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[g][1] == r[g + 1][0] for g in range(world - 1))
            assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= -(-n // world)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    from vqvdb_amd import sharding, synth

    class FakeCodec:  # oracle-backed stand-in for HipCodec
        def __init__(self):
            self.o = Oracle(synth.make_weights(0), [t[0] for t in synth.TENSORS])

        def encode(self, x):
            return self.o.encode(x, threads=2)

        def decode(self, i):
            return self.o.decode(i, threads=2)

    codec = FakeCodec()
    leaves = synth.make_leaves(37, seed=5)           # odd size: ragged shards
    loc = sharding.encode_shard(codec, leaves, rank, world)
    full = sharding.gather_to_rank0(loc, len(leaves), rank, world)
    dt = sharding.max_over_ranks(0.5 + rank)
    assert abs(dt - (0.5 + world - 1)) < 1e-9
    if rank == 0:
        want = codec.encode(leaves)
        rec_full = codec.decode(want)
    else:
        want = rec_full = None
    rloc = sharding.decode_shard(codec, codec.encode(leaves), rank, world)
    rec = sharding.gather_to_rank0(rloc, len(leaves), rank, world)
    if rank == 0:
        q.put((bool(np.array_equal(full, want)), bool(np.array_equal(rec, rec_full))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharded_roundtrip():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=10) == (True, True)
