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


def test_bench_n_gt_1_protocol_and_json_shape_without_a_gpu():
    """bench.py's N > 1 path launched exactly like the driver launches it (torch.distributed.run, one process per rank) with the
    CPU stand-in codec (VQ_BENCH_CPU_REHEARSAL=1, gloo): process group, ranks_seen via an all-reduce, per-rank gathers,
    max-over-ranks timing, the configs[3] workload rule (>= 128 batches per rank), ONE JSON line from rank 0."""
    import json
    import subprocess
    env = dict(os.environ, VQ_BENCH_CPU_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29300 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "collective_backend", "ranks_seen", "devices_seen", "per_rank"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["collective_backend"] == "gloo" and d["scaling"] == "weak"
    assert d["steps"] == 128 and d["config"]["steps_requested"] == 5 and d["config"]["leaves_per_gpu"] == 128 * 65536
    assert "configs[3]" in d["config"]["workload"] and "REHEARSAL" in d["data"]
    assert [o["rank"] for o in d["devices_seen"]] == [0, 1]
    assert len(d["per_rank"]["encode_leaves_per_s"]) == 2 and len(d["per_rank"]["decode_leaves_per_s"]) == 2
    # value = all ranks' leaves over the slowest rank's time: never more than the sum of the per-rank rates
    assert 0 < d["value"] <= sum(d["per_rank"]["encode_leaves_per_s"]) * 1.001
    assert abs(d["ms_per_step"] * d["steps"] * 1e-3 * d["value"] - 2 * 128 * 65536) < 1e-3 * 2 * 128 * 65536
    assert 0 < d["roofline"]["frac"] <= 1 and d["roofline"]["unit"] == "TFLOP/s" and d["roofline"]["bound"] == "mfma"
    assert d["host_path"] and "skipped" in d["host_path"]


def test_bench_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` as ONE plain process (no torch.distributed.run, no WORLD_SIZE in the environment — the shape of
    the driver's N = 1 command): bench.py starts its two ranks itself, exactly one JSON line reaches the original stdout, the
    exit code is the launcher's."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["VQ_BENCH_CPU_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["collective_backend"] == "gloo" and d["steps"] == 128
    assert [o["rank"] for o in d["devices_seen"]] == [0, 1]
    # a rank that fails takes the command down with a non-zero exit code and no JSON line
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--bogus-flag"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]


def test_bench_eight_rank_rehearsal_matches_what_the_first_8_gpu_run_must_print():
    """`bench.py --gpus 8` as the driver launches it, eight gloo ranks with the CPU stand-in codec: ranks_seen == 8, eight distinct
    LOCAL_RANK -> device bindings, the ranks' CPU sets are a partition of this box's allowed cores (or no binding at all when there are
    fewer cores than ranks), every rank covers its configs[3] shard (128 batches = 8 Mi leaves), and the JSON stays ONE line below 64 KB."""
    import json
    import subprocess
    env = dict(os.environ, VQ_BENCH_CPU_REHEARSAL="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 65536, (len(lines), len(lines[0]) if lines else 0)
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["n_devices_seen"] == 8 and d["scaling"] == "weak"
    assert d["steps"] == 128 and d["config"]["leaves_per_gpu"] == 8 * 1024 * 1024 and "configs[3]: 8xMI355X" in d["config"]["workload"]
    seen = d["devices_seen"]
    assert [o["rank"] for o in seen] == list(range(8)) and sorted(o["device_by_local_rank"] for o in seen) == list(range(8))
    allowed = sorted(os.sched_getaffinity(0))
    sets = [o["cpu_set"] for o in seen]
    if len(allowed) >= 8:
        assert all(sets) and sorted(sum(sets, [])) == allowed[:len(allowed) // 8 * 8] and len({tuple(x) for x in sets}) == 8
    else:
        assert not any(sets)
    assert len(d["per_rank"]["encode_leaves_per_s"]) == 8
    assert 0 < d["value"] <= sum(d["per_rank"]["encode_leaves_per_s"]) * 1.001
    assert abs(d["ms_per_step"] * d["steps"] * 1e-3 * d["value"] - 8 * 128 * 65536) < 1e-3 * 8 * 128 * 65536
    assert d["roofline"]["decode_value"] > 0 and d["roofline"]["decode_kernel"]
