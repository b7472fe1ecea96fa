"""Codebook (EMA) training — CPU suite: the oracle's training restatement against golden vectors of the IMPORTED
reference quantizer in training mode (tests/golden/make_golden_train.py), the host logic of
vqvdb_amd/codebook_training.py, and the 2-rank all-reduce path on gloo."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vqvdb_amd import synth
from vqvdb_amd.codebook_training import K, D, STATS_FLOATS, dead_code_reset, metrics_from_stats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-5   # relative to the tensor's scale, like the voxel bar


@pytest.fixture(scope="module")
def gt():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_train_v1.npz"))


def _rel(a, b):
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / max(float(np.abs(b).max()), 1e-30))


def run_reference_schedule(gt, latent_fn, assign_fn, stats_fn, update_fn, weights):
    """Three EMA steps on synth.make_leaves(64, 4000+s), checked against the reference after every step.  Shared by the
    CPU (oracle) and GPU (HIP) tests: the callables hide which implementation runs."""
    state = {"embedding": weights["quantizer.embedding"].copy(), "cluster_size": np.ones(K, np.float32),
             "embed_avg": weights["quantizer.embedding"].copy()}
    for s in range(3):
        z = latent_fn(synth.make_leaves(64, seed=4000 + s), state)
        if s == 0:
            assert _rel(z[:512], gt["z0"]) < TOL
        idx = assign_fn(z, state)
        bad = np.flatnonzero(idx != gt[f"idx{s}"])
        assert all(gt[f"gap{s}"][b] < 1e-3 for b in bad), f"step {s}: assignment differs away from near-ties"
        stats = stats_fn(z, idx, state)
        m = metrics_from_stats(stats, 0.25)
        assert m["rows"] == 4096
        assert abs(m["vq_loss"] - float(gt[f"loss{s}"])) <= TOL * float(gt[f"loss{s}"]) * 10
        assert abs(m["perplexity"] - float(gt[f"ppl{s}"])) <= 1e-4 * float(gt[f"ppl{s}"])
        state = update_fn(stats, state)
        if len(bad) == 0:   # a flipped near-tie moves one row between two codes: buffers then differ legitimately
            for k in ("cluster_size", "embed_avg", "embedding"):
                assert _rel(state[k], gt[f"{k}{s}"]) < TOL, (s, k)
    return state


def test_oracle_training_restatement_matches_reference(oracle, weights, gt):
    run_reference_schedule(
        gt,
        latent_fn=lambda leaves, st: oracle.latent(leaves, threads=8),
        assign_fn=lambda z, st: oracle.vq_assign(z, st["embedding"], threads=8),
        stats_fn=lambda z, idx, st: oracle.vq_stats(z, idx, st["embedding"]),
        update_fn=lambda stats, st: oracle.vq_update(stats, st, 0.95, 1e-4),
        weights=weights)


def test_oracle_stats_are_the_one_hot_products(oracle, weights):
    """encodings_sum / dw / commitment error against a direct numpy evaluation of VQVAE_v2.py:125-137,146."""
    rng = np.random.default_rng(3)
    z = rng.standard_normal((1024, D)).astype(np.float32) * 0.2
    E = weights["quantizer.embedding"]
    idx = oracle.vq_assign(z, E, threads=4)
    d = (z.astype(np.float64) ** 2).sum(1, keepdims=True) + (E.astype(np.float64) ** 2).sum(1) - 2 * z.astype(np.float64) @ E.T.astype(np.float64)
    assert (idx == d.argmin(1)).mean() > 0.999
    stats = oracle.vq_stats(z, idx, E)
    onehot = np.eye(K, dtype=np.float64)[idx]
    assert np.array_equal(stats[:K], onehot.sum(0).astype(np.float32))
    assert _rel(stats[K:K + K * D].reshape(K, D), (onehot.T @ z.astype(np.float64)).astype(np.float32)) < 1e-6
    assert abs(stats[K + K * D:K + K * D + K].sum() - ((z - E[idx]) ** 2).sum()) < 1e-4 * ((z - E[idx]) ** 2).sum()
    assert stats[-1] == 1024 and stats.shape == (STATS_FLOATS,)


def test_dead_code_reset_matches_reference(gt):
    st = {"embedding": torch.from_numpy(gt["embedding2"].copy()), "cluster_size": torch.from_numpy(gt["cluster_size2"].copy()),
          "embed_avg": torch.from_numpy(gt["embed_avg2"].copy())}
    # the reference resets from the last batch's encoder outputs; any flat_z of the same shape draws the same row numbers
    from oracle.oracle import Oracle
    o = Oracle(synth.make_weights(0), [t[0] for t in synth.TENSORS])
    z = torch.from_numpy(o.latent(synth.make_leaves(64, seed=4002), threads=8))
    torch.manual_seed(123)
    n = dead_code_reset(st, z)
    assert n == int(gt["reset_n_dead"]) == 207
    assert np.array_equal(st["cluster_size"].numpy(), gt["reset_cluster_size"])
    for k in ("embedding", "embed_avg"):
        assert _rel(st[k].numpy(), gt[f"reset_{k}"]) < TOL


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    from vqvdb_amd import codebook_training as ct, sharding
    w = synth.make_weights(0)
    o = Oracle(w, [t[0] for t in synth.TENSORS])
    E = w["quantizer.embedding"]
    leaves = synth.make_leaves(64, seed=77)
    lo, hi = sharding.shard_range(len(leaves), rank, world)
    z = o.latent(leaves[lo:hi], threads=2)
    idx = o.vq_assign(z, E, threads=2)
    stats = torch.from_numpy(o.vq_stats(z, idx, E))
    ct.allreduce_stats(stats)                                   # the product's collective, on gloo
    new = o.vq_update(stats.numpy(), {"embedding": E, "cluster_size": np.ones(K, np.float32), "embed_avg": E})
    # dead-code reset: rank 0 draws, everybody ends with the same buffers
    st = {k: torch.from_numpy(v.copy()) for k, v in new.items()}
    g = torch.Generator().manual_seed(5 + rank)                 # different RNG per rank: only rank 0's may matter
    n_dead = ct.dead_code_reset(st, torch.from_numpy(z), generator=g)
    gathered = [None] * world
    dist.all_gather_object(gathered, (n_dead, st["embedding"].numpy().tobytes(), st["cluster_size"].numpy().tobytes()))
    if rank == 0:
        zf = o.latent(leaves, threads=2)
        full = o.vq_stats(zf, o.vq_assign(zf, E, threads=2), E)
        ok_counts = bool(np.array_equal(stats.numpy()[:K], full[:K])) and stats.numpy()[-1] == full[-1] == 4096
        err = float(np.abs(stats.numpy() - full).max() / np.abs(full).max())
        same = all(g_ == gathered[0] for g_ in gathered)
        q.put((ok_counts, err, same, n_dead))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_of_codebook_statistics_gloo():
    """Sharded statistics + all-reduce = statistics of the whole batch (counts exact, sums to fp32 rounding); the
    dead-code reset leaves every rank with rank 0's draw."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_counts, err, same, n_dead = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_counts and err < 1e-6 and same and n_dead > 0


def test_oracle_validation_forward_matches_reference(oracle, weights, gt):
    """training.py:183-199 in eval mode: losses of decode(assign(encoder(x))) against the imported reference's forward."""
    x = synth.make_leaves(64, seed=4100)
    E = weights["quantizer.embedding"]
    z = oracle.latent(x, threads=8)
    idx = oracle.vq_assign(z, E, threads=8)
    m = metrics_from_stats(oracle.vq_stats(z, idx, E), 0.25)
    rec = oracle.decode(idx.reshape(64, 64), threads=8)
    d = rec.astype(np.float64) - x
    assert abs((d * d).mean() - float(gt["eval_recon_mse"])) < 1e-5 * float(gt["eval_recon_mse"])
    assert abs(np.abs(d).mean() - float(gt["eval_recon_l1"])) < 1e-5 * float(gt["eval_recon_l1"])
    assert abs(m["vq_loss"] - float(gt["eval_vq_loss"])) < 1e-4 * float(gt["eval_vq_loss"])
    assert abs(m["perplexity"] - float(gt["eval_ppl"])) < 1e-4 * float(gt["eval_ppl"])
