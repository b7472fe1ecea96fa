"""GPU tests of the codebook-training entry points (vqhip_train_*, run with -m gpu on an MI355X):
HIP kernels vs the oracle's training restatement (bit-exact: same operation order) and vs golden vectors of the
imported reference quantizer in training mode (1e-5)."""
import os

import numpy as np
import pytest
import torch

from oracle.oracle import Oracle, VQ_STATS_FLOATS
from test_codebook_training import gt, run_reference_schedule  # noqa: F401  (fixture + shared schedule)
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codebook_training import CodebookTrainer, K, metrics_from_stats
from vqvdb_amd.codec import HipCodec

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.where(a == 0, 0.0, a).astype(np.float32).view(np.uint32)


@pytest.fixture()
def tcodec(weights):
    c = HipCodec(weightpack.dumps(weights))
    c.train_begin()
    yield c
    c.close()


def _gpu_stats(codec, leaves):
    n = len(leaves)
    x = torch.from_numpy(leaves).cuda()
    stats = torch.zeros(VQ_STATS_FLOATS, device="cuda")
    idx = torch.zeros((n, 64), dtype=torch.uint8, device="cuda")
    z = torch.zeros((n * 64, 128), device="cuda")
    codec.train_vq_stats_device(x.data_ptr(), n, stats.data_ptr(), idx.data_ptr(), z.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return stats, idx.cpu().numpy().reshape(-1), z.cpu().numpy()


def test_training_step_bit_exact_vs_oracle(tcodec, oracle, weights):
    state = {"embedding": weights["quantizer.embedding"].copy(), "cluster_size": np.ones(K, np.float32),
             "embed_avg": weights["quantizer.embedding"].copy()}
    for step, n in enumerate((1000, 33, 1)):   # ragged tiles, 8 row segments (last one short); steps 1, 2 use the LIVE codebook
        leaves = synth.make_leaves(n, seed=31 + step)
        stats, idx, z = _gpu_stats(tcodec, leaves)
        oz = oracle.latent(leaves, threads=8)
        assert np.array_equal(_bits(z), _bits(oz)), step
        oidx = oracle.vq_assign(oz, state["embedding"], threads=8)
        assert np.array_equal(idx, oidx), step
        ostats = oracle.vq_stats(oz, oidx, state["embedding"])
        assert np.array_equal(_bits(stats.cpu().numpy()), _bits(ostats)), step
        tcodec.train_vq_update_device(stats.data_ptr(), 0.95, 1e-4, torch.cuda.current_stream().cuda_stream)
        state = oracle.vq_update(ostats, state, 0.95, 1e-4)
        got = tcodec.train_get_state()
        for k in state:
            assert np.array_equal(_bits(got[k]), _bits(state[k])), (step, k)


def test_statistics_kernels_agree(weights, monkeypatch):
    """The codebook statistics come from per-(segment, code) member lists (vq_ema_lists_k + vq_ema_gather_k, the default) or from every
    (code, segment) wave scanning the segment's indices (VQHIP_TRAIN_EMA=scan, rounds 1-4): the same sums in the same order, bit for bit —
    on a batch whose codes are badly skewed (sparse leaves: a few codes own thousands of rows of a segment, most own none) and on ragged sizes."""
    batches = [np.concatenate([synth.sparse_leaves(3000, seed=5), synth.make_leaves(2003, seed=6)]), synth.make_leaves(129, seed=7), synth.make_leaves(1, seed=8)]
    got = {}
    for mode in ("lists", "scan"):
        monkeypatch.setenv("VQHIP_TRAIN_EMA", mode)
        c = HipCodec(weightpack.dumps(weights))
        c.train_begin()
        got[mode] = [_gpu_stats(c, b)[0].cpu().numpy() for b in batches]
        c.close()
    for a, b in zip(got["lists"], got["scan"]):
        assert a[:K].sum() > 0 and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_training_matches_reference_golden(tcodec, weights, gt):  # noqa: F811
    keep = {}

    def latent(leaves, st):
        keep["stats"], keep["idx"], z = _gpu_stats(tcodec, leaves)
        return z

    def update(stats, st):
        tcodec.train_vq_update_device(keep["stats"].data_ptr(), 0.95, 1e-4, torch.cuda.current_stream().cuda_stream)
        return tcodec.train_get_state()

    run_reference_schedule(gt, latent, lambda z, st: keep["idx"], lambda z, idx, st: keep["stats"].cpu().numpy(), update, weights)


def test_commit_refreshes_inference_tables(tcodec, weights):
    """After training the inference entry points follow the live codebook: same results as a fresh codec / the oracle
    built from the updated embedding."""
    tr = CodebookTrainer(tcodec)
    for s in range(3):
        m = tr.step(torch.from_numpy(synth.make_leaves(256, seed=60 + s)).cuda(), keep_latent=(s == 2))
        assert m["rows"] == 256 * 64 and 0 < m["vq_loss"] < 1 and 1 <= m["perplexity"] <= 256
    sd = tr.state_dict()
    assert set(sd) == {"quantizer.embedding", "quantizer.cluster_size", "quantizer.embed_avg"}
    assert not np.array_equal(sd["quantizer.embedding"], weights["quantizer.embedding"])
    w2 = dict(weights)
    w2["quantizer.embedding"] = sd["quantizer.embedding"]
    o2 = Oracle(w2, [t[0] for t in synth.TENSORS])
    leaves = synth.make_leaves(300, seed=70)
    idx = tcodec.encode(leaves)                                  # refreshes the folded tables on demand
    assert np.array_equal(idx, o2.encode(leaves, threads=8))
    assert np.array_equal(_bits(tcodec.decode(idx)), _bits(o2.decode(idx, threads=8)))
    fresh = HipCodec(weightpack.dumps(w2))
    assert np.array_equal(fresh.encode(leaves), idx)
    fresh.close()
    # dead-code reset through the trainer: dead codes get rows of the kept latent, live state stays consistent
    n_dead = tr.reset_dead_codes(generator=torch.Generator(device="cuda").manual_seed(1))
    st = tcodec.train_get_state()
    assert n_dead == int((sd["quantizer.cluster_size"] < 1.0).sum()) and (st["cluster_size"] >= 1.0).all()
    tr.finish()
    assert np.array_equal(tcodec.encode(leaves[:10]), Oracle({**w2, "quantizer.embedding": st["embedding"]}, [t[0] for t in synth.TENSORS]).encode(leaves[:10]))


def test_training_entry_points_fail_loudly(weights):
    c = HipCodec(weightpack.dumps(weights))
    x = torch.zeros((4, 512), device="cuda")
    stats = torch.zeros(VQ_STATS_FLOATS, device="cuda")
    with pytest.raises(RuntimeError, match="vqhip_train_begin"):
        c.train_vq_stats_device(x.data_ptr(), 4, stats.data_ptr())
    c.train_begin()
    with pytest.raises(RuntimeError, match="null pointer"):
        c.train_vq_stats_device(0, 4, stats.data_ptr())
    c.set_chunk_leaves(32)
    with pytest.raises(RuntimeError, match="chunk size"):
        c.train_vq_stats_device(x.data_ptr(), 64, stats.data_ptr())
    with pytest.raises(RuntimeError, match="decay"):
        c.train_vq_update_device(stats.data_ptr(), 1.5, 1e-4)
    c.close()


def test_validation_forward_matches_reference(tcodec, gt):  # noqa: F811
    tr = CodebookTrainer(tcodec)
    m = tr.evaluate(torch.from_numpy(synth.make_leaves(64, seed=4100)).cuda())
    assert abs(m["recon_mse"] - float(gt["eval_recon_mse"])) < 1e-5 * float(gt["eval_recon_mse"])
    assert abs(m["recon_l1"] - float(gt["eval_recon_l1"])) < 1e-5 * float(gt["eval_recon_l1"])
    assert abs(m["vq_loss"] - float(gt["eval_vq_loss"])) < 1e-4 * float(gt["eval_vq_loss"])
    assert abs(m["perplexity"] - float(gt["eval_ppl"])) < 1e-4 * float(gt["eval_ppl"])
    assert abs(m["recon_error"] - (0.8 * m["recon_mse"] + 0.2 * m["recon_l1"])) < 1e-12
    # evaluation leaves the state untouched and works after training steps too (stale tables refreshed on demand)
    before = tcodec.train_get_state()
    tr.step(torch.from_numpy(synth.make_leaves(128, seed=5)).cuda())
    m2 = tr.evaluate(torch.from_numpy(synth.make_leaves(64, seed=4100)).cuda())
    after = tcodec.train_get_state()
    assert not np.array_equal(before["embedding"], after["embedding"]) and m2["vq_loss"] != m["vq_loss"]
    m3 = tr.evaluate(torch.from_numpy(synth.make_leaves(64, seed=4100)).cuda())
    assert m3 == m2 and all(np.array_equal(after[k], tcodec.train_get_state()[k]) for k in after)


def test_checkpoint_resume_is_bit_identical(weights):
    """state_dict -> a fresh handle -> load_state_dict continues exactly like the uninterrupted run (SURVEY §5 checkpoint row)."""
    batches = [torch.from_numpy(synth.make_leaves(300, seed=80 + s)).cuda() for s in range(3)]
    a = HipCodec(weightpack.dumps(weights))
    ta = CodebookTrainer(a)
    for b in batches:
        ta.step(b)
    b_codec = HipCodec(weightpack.dumps(weights))
    tb = CodebookTrainer(b_codec)
    for b in batches[:2]:
        tb.step(b)
    sd = {k: v.copy() for k, v in tb.state_dict().items()}
    b_codec.close()
    c = HipCodec(weightpack.dumps(weights))          # pack still holds the ORIGINAL codebook: everything comes from the checkpoint
    tc = CodebookTrainer(c)
    tc.load_state_dict(sd)
    m = tc.step(batches[2])
    want, got = ta.state_dict(), tc.state_dict()
    for k in want:
        assert np.array_equal(_bits(want[k]), _bits(got[k])), k
    assert m["rows"] == 300 * 64
    leaves = synth.make_leaves(50, seed=3)
    assert np.array_equal(a.encode(leaves), c.encode(leaves))
    a.close(), c.close()


def test_epoch_driver_trains_validates_and_checkpoints(weights, tmp_path):
    """vqvdb_amd.train_codebook: two short epochs on synthetic leaves; the loop shape of training.py (train pass, validation,
    best-val checkpoint, final save) and a codebook that fits the data better after training."""
    from vqvdb_amd import train_codebook
    (tmp_path / "m.vqw").write_bytes(weightpack.dumps(weights))
    out = train_codebook.main(["train", "--pack", str(tmp_path / "m.vqw"), "--epochs", "2", "--batch_size", "1024", "--leaves_per_epoch", "16384",
                               "--model_path", str(tmp_path / "ckpt" / "quantizer.npz"), "--log_every", "4"])
    h = out["history"]
    assert len(h) == 2 and out["steps_per_epoch"] == 16 and h[0]["leaves_per_s"] > 0
    assert h[1]["train_vq_loss"] < h[0]["train_vq_loss"] and h[1]["val_vq_loss"] < 0.02
    best = np.load(tmp_path / "ckpt" / "quantizer.npz")
    final = np.load(tmp_path / "ckpt" / "quantizer_final.npz")
    for k in ("quantizer.embedding", "quantizer.cluster_size", "quantizer.embed_avg"):
        assert best[k].shape == final[k].shape
    assert not np.array_equal(final["quantizer.embedding"], weights["quantizer.embedding"])
    # the saved codebook drops into a weight pack for inference
    w2 = dict(weights)
    w2["quantizer.embedding"] = final["quantizer.embedding"]
    c = HipCodec(weightpack.dumps(w2))
    leaves = synth.make_leaves(40, seed=1)
    assert np.array_equal(c.encode(leaves), Oracle(w2, [t[0] for t in synth.TENSORS]).encode(leaves, threads=8))
    c.close()


def test_epoch_driver_two_rank_rehearsal(weights, tmp_path):
    """The data-parallel loop of vqvdb_amd.train_codebook with two ranks sharing this box's GPU (gloo): both ranks apply the
    all-reduced statistics, so the rank-0 checkpoint equals a single-rank run over the same global batches up to the
    all-reduce's summation order."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "m.vqw").write_bytes(weightpack.dumps(weights))
    common = ["train", "--pack", str(tmp_path / "m.vqw"), "--epochs", "1", "--leaves_per_epoch", "16384", "--log_every", "2"]
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29700 + os.getpid() % 200), "-m", "vqvdb_amd.train_codebook", *common, "--batch_size", "512",
                        "--backend", "gloo", "--single_gpu_rehearsal", "--model_path", str(tmp_path / "two.npz")],
                       capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "2 rank(s) x batch 512" in r.stdout and "Training completed!" in r.stdout
    from vqvdb_amd import train_codebook
    train_codebook.main([*common, "--batch_size", "1024", "--model_path", str(tmp_path / "one.npz")])
    one, two = np.load(tmp_path / "one_final.npz"), np.load(tmp_path / "two_final.npz")
    for k in ("quantizer.embedding", "quantizer.cluster_size", "quantizer.embed_avg"):
        err = np.abs(one[k].astype(np.float64) - two[k]).max() / np.abs(one[k]).max()
        assert err < 1e-5, (k, err)
