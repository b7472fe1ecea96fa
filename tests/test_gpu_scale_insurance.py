"""SCALE-run insurance: no round has had a second GPU, so the exact code the first multi-GPU bench takes is
executed on the one GPU there is, every round: bench.py's N > 1 path (RCCL process group, the N > 1 JSON shape, the 128-step shard)
with one rank, and the in-process multi-device entry points with device lists that go wrong in the middle."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_distributed_path_with_one_rccl_rank():
    """`python bench.py --gpus 1` with VQ_BENCH_FORCE_DIST=1: init_process_group("nccl") (= RCCL), barriers and max-over-ranks timing,
    all_gather_object of the per-rank rates, the training legs' all-reduces — one rank, 128 steps like a configs[3] shard."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env.update(VQ_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               LOCAL_WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "128", "--warmup", "2", "--no-host-path", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines          # RCCL's banner and everything else went to stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 128 and d["warmup"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["collective_backend"] == "nccl (RCCL)" and d["ranks_seen"] == 1 and d["n_devices_seen"] == 1
    assert d["devices_seen"][0]["rank"] == 0 and d["devices_seen"][0]["name"] and d["devices_seen"][0]["device"] == 0
    # the per-rank table and its flat copies (what a summariser that drops nested objects keeps)
    assert len(d["per_rank"]["encode_leaves_per_s"]) == 1 and len(d["per_rank"]["decode_leaves_per_s"]) == 1
    assert d["per_rank_encode_min"] == d["per_rank_encode_max"] == d["per_rank"]["encode_leaves_per_s"][0]
    assert d["per_rank_decode_min"] == d["per_rank_decode_max"] == d["per_rank"]["decode_leaves_per_s"][0]
    # value = all ranks' leaves over the slowest rank's time (here: the one rank's, barriers included)
    assert abs(d["ms_per_step"] * d["steps"] * 1e-3 * d["value"] - 128 * 65536) < 1e-3 * 128 * 65536
    assert 3e6 < d["value"] <= d["per_rank"]["encode_leaves_per_s"][0] * 1.001
    assert 3e6 < d["decode_value"] == d["decode"]["value"] and d["decode_ms_per_step"] == d["decode"]["ms_per_step"]
    assert 0 < d["whole_path_frac_useful"] <= d["whole_path_frac"] <= 1 and 0 < d["decode_whole_path_frac_useful"] <= d["decode_whole_path_frac"] <= 1
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline_frac"] == d["roofline"]["frac"] <= 1
    # the decode half of the metric NESTED in `roofline` (the object the driver's record keeps whole; its flat twins survive as names only)
    rf = d["roofline"]
    assert rf["decode_value"] == d["decode_value"] and rf["decode_ms_per_step"] == d["decode_ms_per_step"] and rf["decode_unit"] == "leaves/s"
    assert rf["decode_kernel"] == d["decode"]["roofline"]["kernel"] and 0 < rf["decode_frac"] == d["decode_roofline_frac"] <= 1
    assert 0 < rf["decode_whole_path_frac_useful"] <= rf["decode_whole_path_frac"] <= 1 and 0.5 < rf["dec_tail_ms"] < 3.0
    assert "decode_traffic_bytes" in rf and "decode_workload" in rf
    # the training legs ran their collectives through the same process group
    assert "RCCL" in d["codebook_training"]["collective"] and "1 rank" in d["codebook_training"]["collective"]
    assert "error" not in d["codebook_training"] and "error" not in d["full_training"]
    assert d["full_training"]["per_rank_batch_2048"]["leaves_per_s"] > 0


def test_multi_device_lists_that_go_wrong_in_the_middle():
    """vqhip_multi_create with an invalid ordinal in the middle or at the front of the device list: a clean error (the handles created
    before it are destroyed), and the next creation on the good device works and computes the same results."""
    from vqvdb_amd import synth, weightpack
    from vqvdb_amd.codec import HipCodec, HipMultiCodec
    pack = weightpack.dumps(synth.make_weights(seed=0))
    for bad in ([0, 99, 0], [99, 0], [0, 0, -1], [0, 7, 0, 0]):   # (a 1-GPU box: every ordinal but 0 is invalid, 7 included)
        with pytest.raises(RuntimeError, match="device_id out of range"):
            HipMultiCodec(pack, bad)
    with pytest.raises((RuntimeError, ValueError)):
        HipMultiCodec(pack, [])
    leaves = synth.make_leaves(777, seed=41)
    ref = HipCodec(pack)
    want = ref.encode(leaves)
    m = HipMultiCodec(pack, [0, 0])
    assert np.array_equal(m.encode(leaves), want)
    assert np.array_equal(m.decode(want).view(np.uint32), ref.decode(want).view(np.uint32))
    assert [d for d, _, _ in m.worker_info()] == [0, 0]
    m.close()
    ref.close()


def test_eight_persistent_workers_on_one_device():
    """vqhip_multi_create with eight entries (all device 0 here): eight persistent workers, ranges of a 1/8-scale configs[3] job
    (8 x 128 Ki leaves instead of 8 x 8 Mi), results identical to one handle; the shape the first 8-GPU box will run with eight ordinals."""
    from vqvdb_amd import synth, weightpack
    from vqvdb_amd.codec import HipCodec, HipMultiCodec
    pack = weightpack.dumps(synth.make_weights(seed=0))
    base = synth.make_leaves(4096, seed=88)
    n = 8 * 16384 + 5                      # ragged: the last range is 5 leaves longer
    leaves = np.tile(base, (-(-n // len(base)), 1))[:n]
    ref = HipCodec(pack)
    want_idx = ref.encode(leaves[:4096])
    want_rec = ref.decode(want_idx)
    ref.close()
    m = HipMultiCodec(pack, [0] * 8)
    assert [d for d, _, _ in m.worker_info()] == [0] * 8
    for _ in range(2):
        idx = m.encode(leaves)
        assert idx.shape == (n, 64)
        for k in range(0, n - 4095, 4096):
            assert np.array_equal(idx[k:k + 4096], want_idx), k
        assert np.array_equal(idx[-5:], want_idx[:5])
        rec = m.decode(idx)
        assert np.array_equal(rec[:4096].view(np.uint32), want_rec.view(np.uint32)) and np.array_equal(rec[-4096 - 5:-5].view(np.uint32), want_rec.view(np.uint32))
    m.close()
