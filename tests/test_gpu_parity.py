"""GPU parity tests (run with -m gpu on an MI355X): HIP path through the C ABI vs the CPU oracle
and vs the golden vectors of the imported reference.

Bars (BASELINE.json north_star): indices bit-exact; voxels within 1e-5 relative.  Because the
kernels share the oracle's accumulation orders, voxels and every intermediate are in fact
compared BIT-EXACTLY against the oracle (+0/-0 identified)."""
import numpy as np
import pytest

from conftest import rel_err
from oracle.oracle import DEBUG_SHAPES, DEC_DEBUG, ENC_DEBUG
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import (BackendType, CodecConfig, DataType, HipCodec, IVQVAECodec, TensorView)

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _bits(a):
    return np.where(a == 0, 0.0, a).astype(np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def pack(weights):
    return weightpack.dumps(weights)


@pytest.fixture(scope="module")
def codec(pack):
    c = HipCodec(pack)
    yield c
    c.close()


def test_native_library_loaded_and_mfma_is_fmaf_chain(codec):
    # the kernels' bit-exactness contract rests on this hardware property (cdna guide §3)
    assert codec.selftest_mfma() == [0, 0]
    assert codec.latent_shape() == [4, 4, 4]


def test_encode_bit_exact_vs_oracle_all_layers(codec, oracle):
    leaves = np.concatenate([synth.make_leaves(120, seed=1234), synth.edge_leaves()])
    codec.debug_enable(True)
    idx = codec.encode(leaves)
    oidx, dbg = oracle.encode(leaves, threads=8, debug=ENC_DEBUG)
    for name in ENC_DEBUG:
        if name in ("e_x12", "e_z"):   # gated activations and the 128-channel latent are never formed on the GPU
            continue
        c, p = DEBUG_SHAPES[name]
        assert np.array_equal(_bits(codec.debug_fetch(name, len(leaves), c, p)), _bits(dbg[name])), name
    codec.debug_enable(False)
    assert np.array_equal(idx, oidx)


def test_decode_bit_exact_vs_oracle_all_layers(codec, oracle, golden):
    idx = np.concatenate([golden["idx_rand"][:100], golden["idx_edge"]])
    codec.debug_enable(True)       # every intermediate at its own address; the one-kernel decoder front then also stores the raw stem output
    rec = codec.decode(idx)
    orec, dbg = oracle.decode(idx, threads=8, debug=DEC_DEBUG)
    for name in ["d_ystem", "d_d2", "d_y4", "d_x6"]:
        c, p = DEBUG_SHAPES[name]
        assert np.array_equal(_bits(codec.debug_fetch(name, len(idx), c, p)), _bits(dbg[name])), name
    codec.debug_enable(False)
    assert np.array_equal(_bits(codec.decode(idx)), _bits(rec))      # the same voxels without the debug stores
    assert np.array_equal(_bits(rec), _bits(orec))
    assert float((np.abs(rec - orec) / np.abs(orec)).max()) < TOL     # the stated tolerance, trivially met


def test_golden_reference_vectors(codec, golden):
    """HIP vs outputs of the imported reference model (tests/golden/make_golden.py)."""
    idx = codec.encode(synth.make_leaves(1024, seed=1234))
    bad = np.nonzero(idx.reshape(-1) != golden["idx_rand"].reshape(-1))[0]
    ties = dict(zip(golden["tie_pos"].tolist(), golden["tie_gap"].tolist()))
    assert all(ties.get(int(p), 1.0) < 1e-5 for p in bad) and len(bad) <= 2      # measured: 0
    idx_e = codec.encode(synth.edge_leaves())
    assert np.array_equal(idx_e, golden["idx_edge"])
    rec = codec.decode(golden["idx_rand"][:64])
    assert float((np.abs(rec - golden["rec_rand"]) / np.abs(golden["rec_rand"])).max()) < TOL
    rec_e = codec.decode(golden["idx_edge"])
    assert float((np.abs(rec_e - golden["rec_edge"]) / np.abs(golden["rec_edge"])).max()) < TOL


@pytest.mark.parametrize("n", [1, 31, 32, 33, 63, 65, 257])
def test_ragged_batches_and_batch_independence(codec, oracle, n):
    leaves = synth.make_leaves(300, seed=7)
    full = codec.encode(leaves)
    part = codec.encode(leaves[:n])
    assert np.array_equal(part, full[:n])
    assert np.array_equal(part, oracle.encode(leaves[:n], threads=8))
    rfull = codec.decode(full)
    rpart = codec.decode(full[:n])
    assert np.array_equal(_bits(rpart), _bits(rfull[:n]))


def test_chunking_is_invisible(pack, oracle):
    c = HipCodec(pack)
    c.set_chunk_leaves(64)
    leaves = synth.make_leaves(200, seed=11)
    idx = c.encode(leaves)
    assert np.array_equal(idx, oracle.encode(leaves, threads=8))
    rec = c.decode(idx)
    assert np.array_equal(_bits(rec), _bits(oracle.decode(idx, threads=8)))
    c.close()


def test_full_size_batch_properties(codec, oracle):
    """BASELINE config sizes (64k-leaf batches) through size-independent properties: a sampled
    subset equals the oracle, duplicates encode identically wherever they sit in the batch,
    decode(encode(x)) is idempotent under re-encode->decode of identical indices."""
    n = 65536
    base = synth.make_leaves(4096, seed=5)
    leaves = np.tile(base, (n // 4096, 1))
    idx = codec.encode(leaves)
    assert np.array_equal(idx[:4096], idx[-4096:]) and np.array_equal(idx[:4096], idx[8 * 4096: 9 * 4096])
    sample = np.arange(0, 4096, 37)
    assert np.array_equal(idx[sample], oracle.encode(base[sample], threads=8))
    rec = codec.decode(idx)
    assert np.array_equal(_bits(rec[:4096]), _bits(rec[-4096:]))
    assert np.array_equal(_bits(rec[sample[:32]]), _bits(oracle.decode(idx[sample[:32]], threads=8)))
    assert np.isfinite(rec).all() and rec.min() > 0.0 and rec.max() < 1.0   # sigmoid range
    hist = np.bincount(idx.reshape(-1), minlength=256)
    assert hist.sum() == n * 64 and (hist > 0).sum() > 64


def test_interface_mirror_matches_reference_behaviour(pack, oracle, capsys):
    cfg = CodecConfig(device=CodecConfig.Device.CUDA, source=pack)
    be = IVQVAECodec.create(cfg, BackendType.HIP)
    assert be is not None and be.getLatentShape() == [4, 4, 4]
    leaves = synth.make_leaves(65, seed=3)
    t = be.encode(TensorView(leaves, [65, 1, 8, 8, 8], DataType.FLOAT32))
    assert t.shape == [65, 4, 4, 4] and t.dtype == DataType.UINT8
    assert np.array_equal(t.getData().reshape(65, 64), oracle.encode(leaves, threads=8))
    r = be.decode(TensorView(t.getData(), t.shape, DataType.UINT8))
    assert r.shape == [65, 1, 8, 8, 8] and r.dtype == DataType.FLOAT32       # 5-D like the reference
    with pytest.raises(RuntimeError, match="encode expects FLOAT32 data."):
        be.encode(TensorView(leaves, [65, 1, 8, 8, 8], DataType.UINT8))
    with pytest.raises(RuntimeError, match="decode expects UINT8 data."):
        be.decode(TensorView(leaves, [65, 4, 4, 4], DataType.FLOAT32))
    # creation failures never raise: nullptr (None) + a message on stderr (IVQVAECodec.cpp:106-109)
    assert IVQVAECodec.create(CodecConfig(device=CodecConfig.Device.CUDA, source="/nonexistent.vqw"), BackendType.HIP) is None
    assert "Model file not found" in capsys.readouterr().err
    assert IVQVAECodec.create(CodecConfig(device=CodecConfig.Device.CPU, source=pack), BackendType.HIP) is None
    assert IVQVAECodec.create(cfg, BackendType.ONNX) is None


def test_malformed_weight_packs_fail_loudly(pack, weights):
    with pytest.raises(RuntimeError, match="bad magic"):
        HipCodec(b"NOTAPACK" + pack[8:])
    w = dict(weights)
    del w["decoder.final.bias"]
    with pytest.raises(RuntimeError, match="missing tensor 'decoder.final.bias'"):
        HipCodec(weightpack.dumps(w))
    w = dict(weights)
    w["quantizer.embedding"] = w["quantizer.embedding"][:128]
    with pytest.raises(RuntimeError, match="unexpected shape"):
        HipCodec(weightpack.dumps(w))


def test_cpp_adapter_orchestrator_roundtrip(codec, pack, tmp_path):
    """C++ host side (HipBackend : IVQVAECodec via IVQVAECodec::create, orchestrator-style batching,
    .vqvdb framing) gives the same bytes as the C-ABI path, for SOP-default and large batches."""
    import subprocess
    from vqvdb_amd.build import build_harness
    HARNESS = build_harness()
    leaves = synth.make_leaves(1000, seed=21)
    (tmp_path / "m.vqw").write_bytes(pack)
    leaves.tofile(tmp_path / "in.f32")
    want_idx = codec.encode(leaves)
    want_rec = codec.decode(want_idx)
    for batch in (64, 333, 4096):
        r = subprocess.run([HARNESS, "compress", str(tmp_path / "m.vqw"), str(tmp_path / "in.f32"), str(tmp_path / "o.vqvdb"), str(batch)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        raw = (tmp_path / "o.vqvdb").read_bytes()
        body = np.frombuffer(raw[-1000 * 76:], dtype=np.uint8).reshape(1000, 76)
        assert np.array_equal(body[:, 12:], want_idx)
        r = subprocess.run([HARNESS, "decompress", str(tmp_path / "m.vqw"), str(tmp_path / "o.vqvdb"), str(tmp_path / "out.f32"), str(batch)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got = np.fromfile(tmp_path / "out.f32", dtype=np.float32).reshape(1000, 512)
        assert np.array_equal(_bits(got), _bits(want_rec))
    r = subprocess.run([HARNESS, "errors", str(tmp_path / "m.vqw")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_embedded_model_harness_matches_path_source(codec, pack, tmp_path):
    """CodecConfig::source = EmbeddedModel{} (both reference SOPs: SOP_VQVDB_Encoder.cpp:63-67, SOP_VQVDB_Decoder.cpp:58-62):
    leaf_harness_embedded (the weight pack compiled in with `python -m vqvdb_amd.weightpack --header`) round-trips
    5 000 leaves through HipBackend and gives the bytes of the path-source build and of the C-ABI path."""
    import subprocess
    from vqvdb_amd.build import build_default_embedded_harness, build_harness
    emb, plain = build_default_embedded_harness(), build_harness()
    assert pack == weightpack.dumps(synth.make_weights(0)), "the embedded harness carries the seed-0 pack"
    leaves = synth.make_leaves(5000, seed=77)
    (tmp_path / "m.vqw").write_bytes(pack)
    leaves.tofile(tmp_path / "in.f32")
    want_idx = codec.encode(leaves)
    want_rec = codec.decode(want_idx)
    files = {}
    for tag, exe, src in (("emb", emb, "@embedded"), ("path", plain, str(tmp_path / "m.vqw"))):
        for batch in (64, 2048):
            r = subprocess.run([exe, "compress", src, str(tmp_path / "in.f32"), str(tmp_path / f"{tag}.vqvdb"), str(batch)], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            files[tag, batch] = (tmp_path / f"{tag}.vqvdb").read_bytes()
            r = subprocess.run([exe, "decompress", src, str(tmp_path / f"{tag}.vqvdb"), str(tmp_path / f"{tag}.f32"), str(batch)], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            got = np.fromfile(tmp_path / f"{tag}.f32", dtype=np.float32).reshape(5000, 512)
            assert np.array_equal(_bits(got), _bits(want_rec)), (tag, batch)
    assert len(set(files.values())) == 1
    body = np.frombuffer(files["emb", 64][-5000 * 76:], dtype=np.uint8).reshape(5000, 76)
    assert np.array_equal(body[:, 12:], want_idx)
    r = subprocess.run([emb, "errors", "@embedded"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([plain, "errors", "@embedded"], capture_output=True, text=True)       # no pack in that build: refused, loudly
    assert r.returncode == 1 and "no weight pack given" in r.stderr


def _origins(n, start=0):
    i = np.arange(start, start + n, dtype=np.int64)
    return np.stack([8 * (i % 1024), 8 * ((i // 1024) % 1024), 8 * (i // 1048576)], axis=1).astype(np.int32)


def test_stream_file_entry_points_match_block_path(codec, tmp_path):
    """vqhip_compress_file / vqhip_decompress_file (file I/O || GPU || leaf insert overlapped inside the library)
    produce byte-for-byte the .vqvdb the block path + host framing produces, and the same voxels; multi-grid,
    ragged last batch, per-leaf buffers, empty grid."""
    from vqvdb_amd import vqvdbfile
    la, lb = synth.make_leaves(10000, seed=5), synth.make_leaves(333, seed=6)
    oa, ob = _origins(10000), _origins(333, start=5_000_000)
    tr = np.arange(16, dtype=np.float32) * 0.25
    ia, ib = codec.encode(la), codec.encode(lb)
    want = vqvdbfile.dumps([vqvdbfile.Grid("density", oa, ia), vqvdbfile.Grid("temperature", ob, ib, tr),
                            vqvdbfile.Grid("empty", np.zeros((0, 3), np.int32), np.zeros((0, 64), np.uint8))])
    path = tmp_path / "s.vqvdb"
    for batch in (0, 4096, 1000):
        st = codec.compress_file(path, [("density", oa, la, None), ("temperature", ob, [lb[i].copy() for i in range(333)], tr),
                                        ("empty", np.zeros((0, 3), np.int32), np.zeros((0, 512), np.float32), None)], batch_leaves=batch)
        assert st["leaves"] == 10333 and st["grids"] == 3
        assert path.read_bytes() == want
        grids, st = codec.decompress_file(path, batch_leaves=batch)
        assert st["leaves"] == 10333 and [g[0] for g in grids] == ["density", "temperature", "empty"]
        assert np.array_equal(grids[1][1], tr) and np.array_equal(grids[0][1], np.eye(4, dtype=np.float32).reshape(16))
        assert np.array_equal(grids[0][2], oa) and np.array_equal(grids[1][2], ob) and len(grids[2][2]) == 0
        assert np.array_equal(_bits(grids[0][3]), _bits(codec.decode(ia)))
        assert np.array_equal(_bits(grids[1][3]), _bits(codec.decode(ib)))


def test_stream_file_errors_fail_loudly(codec, tmp_path):
    from vqvdb_amd import vqvdbfile
    idx = codec.encode(synth.make_leaves(100, seed=9))
    good = vqvdbfile.dumps([vqvdbfile.Grid("density", _origins(100), idx)])
    cases = {"trunc": (good[:-30], "File truncated"), "magic": (b"NOTVQ" + good[5:], "Invalid file magic"),
             "version": (good[:5] + b"\x02" + good[6:], "Unsupported .vqvdb version"),
             "shape": (good[:12 + 4 + 7 + 64] + b"\x08\x00\x08\x00\x08\x00" + good[12 + 4 + 7 + 64 + 6:], "latent shape"),
             "header": (good[:7], "Failed to read file header")}
    for name, (blob, msg) in cases.items():
        p = tmp_path / f"{name}.vqvdb"
        p.write_bytes(blob)
        with pytest.raises(RuntimeError, match=msg):
            codec.decompress_file(p)
    with pytest.raises(RuntimeError, match="Cannot open input file"):
        codec.decompress_file(tmp_path / "missing.vqvdb")
    with pytest.raises(RuntimeError, match="Cannot open output file"):
        codec.compress_file(tmp_path / "no_such_dir" / "x.vqvdb", [("g", _origins(1), synth.make_leaves(1), None)])
    with pytest.raises(RuntimeError, match="1..255 grids"):
        codec.compress_file(tmp_path / "x.vqvdb", [])
    # the codec is still usable after every failure
    assert np.array_equal(codec.encode(synth.make_leaves(100, seed=9)), idx)


def test_cpp_harness_stream_modes_match_orchestrator_modes(codec, pack, tmp_path):
    """leaf_harness compress_stream / decompress_stream (C ABI whole-file entry points, hash-map leaf store standing in
    for tree.touchLeaf) write the same files as the orchestrator-shaped compress / decompress modes."""
    import subprocess
    from vqvdb_amd.build import build_harness
    HARNESS = build_harness()
    leaves = synth.make_leaves(5000, seed=33)
    (tmp_path / "m.vqw").write_bytes(pack)
    leaves.tofile(tmp_path / "in.f32")
    m, i = str(tmp_path / "m.vqw"), str(tmp_path / "in.f32")
    for mode, out in (("compress", "a.vqvdb"), ("compress_stream", "b.vqvdb")):
        r = subprocess.run([HARNESS, mode, m, i, str(tmp_path / out), "2048"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    assert (tmp_path / "a.vqvdb").read_bytes() == (tmp_path / "b.vqvdb").read_bytes()
    for mode, out in (("decompress", "a.f32"), ("decompress_stream", "b.f32")):
        r = subprocess.run([HARNESS, mode, m, str(tmp_path / "a.vqvdb"), str(tmp_path / out), "2048"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    assert (tmp_path / "a.f32").read_bytes() == (tmp_path / "b.f32").read_bytes()
    assert np.array_equal(_bits(np.fromfile(tmp_path / "b.f32", dtype=np.float32).reshape(5000, 512)), _bits(codec.decode(codec.encode(leaves))))


def test_config1_10k_random_leaves_roundtrip(codec, oracle):
    """BASELINE configs[0] size: 10 000 random leaves, K=256 D=128, encode -> indices -> decode, every
    index and every voxel compared with the CPU oracle."""
    leaves = synth.make_leaves(10000, seed=1234)
    idx = codec.encode(leaves)
    oidx = oracle.encode(leaves, threads=16)
    assert np.array_equal(idx, oidx)
    rec = codec.decode(idx)
    assert np.array_equal(_bits(rec), _bits(oracle.decode(oidx, threads=16)))
    # round trip is a contraction towards the codebook: re-encoding the reconstruction is deterministic
    assert np.array_equal(codec.encode(rec), codec.encode(rec.copy()))


def test_config2_one_million_leaves_permutation_property(codec, oracle):
    """BASELINE configs[1] size: 1 000 000 leaves in 65 536-leaf chunks.  Leaves are independent, so the
    indices of a shuffled multiset must be the shuffled indices of its 4096 distinct members, whose
    indices are in turn checked against the oracle (bit-exact)."""
    base = synth.make_leaves(4096, seed=77)
    base_idx = codec.encode(base)
    assert np.array_equal(base_idx, oracle.encode(base, threads=16))
    rng = np.random.default_rng(5)
    perm = rng.integers(0, 4096, size=1_000_000)
    big = base[perm]
    idx = codec.encode(big)
    assert idx.shape == (1_000_000, 64)
    assert np.array_equal(idx, base_idx[perm])
    rec = codec.decode(idx[:200_000])
    base_rec = codec.decode(base_idx)
    assert np.array_equal(_bits(rec), _bits(base_rec[perm[:200_000]]))


def test_config3_four_million_leaf_file_streamed_decode(codec, oracle, tmp_path):
    """BASELINE configs[2] at full size: a single-grid .vqvdb of 4 Mi leaves decoded by vqhip_decompress_file in 65 536-leaf
    batches (reference path: src/orchestrator/VQVAECodec.cpp:137-208, src/Utils/VQVDB_Reader.cpp:240-335).  Leaves are independent,
    so the file is built from 4096 distinct index rows whose decodes are checked bit-exactly against the oracle; every one of the
    4 Mi decoded leaves must then equal the decode of its row, and the origins must come back in file order."""
    from vqvdb_amd import hostbench
    n = 4 * 1024 * 1024
    rng = np.random.default_rng(2024)
    base_idx = rng.integers(0, 256, size=(4096, 64), dtype=np.uint8)
    base_idx[:1024] = codec.encode(synth.make_leaves(1024, seed=99))          # realistic rows (encoder outputs) among the random ones
    base_rec = codec.decode(base_idx)
    assert np.array_equal(_bits(base_rec), _bits(oracle.decode(base_idx, threads=16)))
    perm = rng.integers(0, 4096, size=n).astype(np.int32)
    path = tmp_path / "c3.vqvdb"
    hostbench.write_index_file(str(path), lambda s, k: base_idx[perm[s:s + k]], n)
    assert path.stat().st_size == 12 + 4 + 7 + 64 + 6 + 4 + 76 * n
    pool = np.empty((n, 512), dtype=np.float32)
    grids, st = codec.decompress_file(path, batch_leaves=65536, out=pool)
    assert st["leaves"] == n and st["grids"] == 1 and len(grids) == 1 and grids[0][0] == "density"
    assert np.array_equal(grids[0][2], hostbench.origins_of(n))               # origins, file order
    assert grids[0][3].shape == (n, 512) and grids[0][3].ctypes.data == pool.ctypes.data
    bad = 0
    for s in range(0, n, 1 << 18):
        bad += int((_bits(pool[s:s + (1 << 18)]) != _bits(base_rec[perm[s:s + (1 << 18)]])).any(axis=1).sum())
    assert bad == 0, f"{bad} of {n} leaves differ from the decode of their index row"
    print(f"config3: {n} leaves in {st['wall_s']:.3f} s = {n / st['wall_s'] / 1e6:.2f} M leaves/s "
          f"(read {st['read_s']:.3f} s, alloc {st['alloc_s']:.3f} s, scatter {st['copy_s']:.3f} s, waited for reader {st['io_wait_s']:.3f} s)")


def test_workspace_is_compact_for_inference_and_full_for_debug(pack, oracle):
    """<= 8 GB of workspace at the default 65 536-leaf chunk (three shared activation regions); debug mode
    switches to the full layout (every intermediate at its own address, sized for the call that needs it) and back-to-back encode / decode on the shared regions
    stay bit-exact."""
    c = HipCodec(pack)
    c.reserve(65536)
    assert c.chunk_leaves() == 65536 and 0 < c.workspace_bytes() < 8e9, c.workspace_bytes()
    compact = c.workspace_bytes()
    leaves = synth.make_leaves(3000, seed=17)
    idx = c.encode(leaves)
    rec = c.decode(idx)
    assert np.array_equal(idx, c.encode(leaves)) and np.array_equal(_bits(rec), _bits(c.decode(idx)))      # regions re-used across calls
    assert np.array_equal(idx[:512], oracle.encode(leaves[:512], threads=16))
    with pytest.raises(RuntimeError, match="compact inference workspace"):
        c.debug_fetch("e_a1", 32, 16, 512)
    c.debug_enable(True)
    assert np.array_equal(c.encode(leaves), idx)
    # the switch to the full layout (every intermediate at its own address, 2.4x the bytes per leaf) is sized for what THIS call
    # needs — 94 tiles here — not for the 65 536 leaves the handle was reserved for (that would be 16 GB for a 3000-leaf debug pass)
    per_leaf_compact, per_leaf_full = compact / 65536, c.workspace_bytes() / (94 * 32)
    assert c.workspace_bytes() < 0.2 * compact and per_leaf_full > 1.8 * per_leaf_compact
    assert c.debug_fetch("e_a1", 32, 16, 512).shape == (32, 16, 512)
    c.debug_enable(False)
    big = np.tile(leaves, (22, 1))[:65536]
    assert np.array_equal(c.encode(big)[:3000], idx)          # grows again (a full layout also serves inference)
    c.close()


def test_leaf_pointer_entry_points(pack):
    """scattered per-leaf buffers in, scattered per-leaf buffers out, across chunk boundaries."""
    c = HipCodec(pack)
    c.set_chunk_leaves(4096)
    leaves = synth.make_leaves(10000, seed=31)
    order = np.random.default_rng(1).permutation(10000)
    pool = leaves[order].copy()                       # leaf i lives at pool[inv[i]] — non-contiguous w.r.t. leaf order
    inv = np.argsort(order)
    want = c.encode(leaves)
    got = c.encode_leaves([pool[inv[i]] for i in range(10000)])
    assert np.array_equal(got, want)
    out_pool = np.zeros_like(pool)
    c.decode_leaves(want, [out_pool[inv[i]] for i in range(10000)])
    assert np.array_equal(_bits(out_pool[inv]), _bits(c.decode(want)))
    c.close()


def test_large_one_chunk_host_calls_are_cut_into_pieces_without_changing_a_bit(pack, monkeypatch):
    """Round 6: a host-memory call that fits one chunk but is large is cut into up to 8 pieces of >= 8192 leaves inside the library (gather,
    H2D, kernels, D2H and scatter of neighbouring pieces overlap).  Block and leaf-pointer entry points, ragged sizes, pieces that are not a
    multiple of a tile: identical bytes to the uncut call (VQHIP_HOST_SPLIT=1) and to a finer cut."""
    n = 8 * 8192 - 37                                  # 7 pieces of 9357 -> rounded up to tiles, ragged last piece
    base = synth.make_leaves(4096, seed=515)
    leaves = np.tile(base, (-(-n // 4096), 1))[:n]
    leaves[::7] *= 0.5                                 # (not all periods identical)
    results = []
    for env in ({"VQHIP_HOST_SPLIT": "1"}, {}, {"VQHIP_HOST_SPLIT": "5,4096"}):
        monkeypatch.delenv("VQHIP_HOST_SPLIT", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = HipCodec(pack)
        idx = c.encode(leaves)
        rec = c.decode(idx)
        idx_p = c.encode_leaves([leaves[i] for i in range(n)])
        out = np.zeros_like(leaves)
        c.decode_leaves(idx, [out[i] for i in range(n)])
        small = c.encode(leaves[:16383])               # below two pieces' worth: never cut
        c.close()
        assert np.array_equal(idx, idx_p) and np.array_equal(_bits(rec), _bits(out)) and np.array_equal(small, idx[:16383])
        results.append((idx, rec))
    for idx, rec in results[1:]:
        assert np.array_equal(idx, results[0][0]) and np.array_equal(_bits(rec), _bits(results[0][1]))


def test_in_process_multi_device_sharding(codec, pack):
    """vqhip_multi_*: leaf ranges over several device handles, one host thread each, results in place.
    Only one GPU is visible here, so the three 'devices' are three independent handles on device 0 —
    that exercises the range split, the threading and the error plumbing, not xGMI."""
    from vqvdb_amd.codec import HipMultiCodec
    m = HipMultiCodec(pack, [0, 0, 0])
    leaves = synth.make_leaves(1000, seed=13)      # 334 + 334 + 332: ragged ranges
    idx = m.encode(leaves)
    assert np.array_equal(idx, codec.encode(leaves))
    assert np.array_equal(_bits(m.decode(idx)), _bits(codec.decode(idx)))
    assert np.array_equal(m.encode(leaves[:2]), idx[:2])       # fewer leaves than devices
    # the per-device host threads are persistent: many calls, alternating directions and sizes, re-use them
    big = synth.make_leaves(9000, seed=14)
    want = codec.encode(big)
    for _ in range(3):
        assert np.array_equal(m.encode(big), want)
        assert np.array_equal(_bits(m.decode(want[:777])), _bits(codec.decode(want[:777])))
    info = m.worker_info()
    assert [d for d, _, _ in info] == [0, 0, 0] and all(nn >= -1 and cb >= 0 for _, nn, cb in info)
    with pytest.raises(RuntimeError, match="device_id out of range"):
        HipMultiCodec(pack, [0, 99])
    m.close()


@pytest.mark.parametrize("n", [1, 64, 100, 300, 1024, 2048, 20000, 40000])
def test_small_batch_split_path_is_bit_identical(pack, oracle, n):
    """Position-split kernels with statistics fused as per-block partials (default policy: passes of <= 1600 (encode) / 1728 (decode) tiles;
    the tiniest batches additionally split the output channels of the 4^3 convs, the folded tail has its own small-batch kernel) against the one-wave-per-tile path and the oracle: indices,
    every stored intermediate and voxels identical."""
    leaves = synth.make_leaves(n, seed=900 + n)
    a, b = HipCodec(pack), HipCodec(pack)
    b.set_small_batch_tiles(0)                      # b: classic path
    a.debug_enable(True), b.debug_enable(True)
    ia, ib = a.encode(leaves), b.encode(leaves)
    assert np.array_equal(ia, ib)
    for name in ("e_y1", "e_a1", "e_y4", "e_a6", "e_x7", "e_y9", "e_x11") if n <= 20000 else ("e_x7", "e_x11"):
        c, p = DEBUG_SHAPES[name]
        assert np.array_equal(_bits(a.debug_fetch(name, n, c, p)), _bits(b.debug_fetch(name, n, c, p))), name
    if n <= 1024:
        assert np.array_equal(ia, oracle.encode(leaves, threads=16))
    assert np.array_equal(_bits(a.decode(ia)), _bits(b.decode(ib)))
    a.close(), b.close()


def test_small_batch_policy_setter(pack):
    """vqhip_set_small_batch_tiles: -1 = automatic choice (default), 0 = never split, n = plain threshold; anything below -1 is
    rejected loudly.  Whatever the policy, results do not change."""
    c = HipCodec(pack)
    x = synth.make_leaves(3000, seed=31)
    ref = c.encode(x)
    for tiles in (0, 1, 50, 1 << 20, -1):
        c.set_small_batch_tiles(tiles)
        assert np.array_equal(c.encode(x), ref), tiles
    with pytest.raises(RuntimeError, match="small_batch_tiles"):
        c.set_small_batch_tiles(-2)
    c.close()


def test_bench_two_rank_code_path_rehearsal():
    """bench.py's N > 1 path (rank-sharded seeds, barriers, max-over-ranks timing, all-reduce of the codebook statistics,
    one JSON line from rank 0) started as the PLAIN command `python bench.py --gpus 2` — bench.py launches its own ranks under
    torch.distributed.run (the torchrun-launched form is covered on CPU by tests/test_sharding_gloo.py) — with two ranks sharing
    the one GPU of this box (gloo instead of RCCL; the throughput of such a run means nothing)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["VQ_BENCH_SINGLE_GPU_REHEARSAL"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["decode"]["value"] > 0
    # N > 1 = BASELINE configs[3]: every rank covers its 8 Mi-leaf shard (128 batches) even when fewer steps were asked for
    assert d["steps"] == 128 and d["config"]["steps_requested"] == 2 and d["config"]["leaves_per_gpu"] == 8 * 1024 * 1024
    assert d["ranks_seen"] == 2 and d["collective_backend"] == "gloo" and len(d["devices_seen"]) == 2
    assert len(d["per_rank"]["encode_leaves_per_s"]) == 2 and d["per_rank"]["encode_min_max"][0] > 0
    assert 0 < d["roofline"]["frac"] <= 1 and 0 < d["decode"]["roofline"]["frac"] <= 1
    assert 0 < d["roofline"]["whole_path_frac"] <= 1 and 0 < d["decode"]["roofline"]["whole_path_frac"] <= 1
    ct = d["codebook_training"]
    assert "error" not in ct and "over 2 rank(s)" in ct["collective"] and ct["per_rank_batch_2048"]["leaves_per_s"] > 0


def test_one_handle_mixed_batch_sizes_no_stale_state(pack, oracle):
    """A long-lived handle (the SOP node cache) sees batches of any size in any order: the split path, the one-wave-per-tile
    path, multi-chunk calls and workspace growth share the statistics / activation buffers and must not leak state."""
    rng = np.random.default_rng(11)
    c = HipCodec(pack)
    c.set_chunk_leaves(4096)                                  # 21000 leaves -> 6 chunks; 4096-leaf chunks take the split path
    sizes = [5, 21000, 64, 3000, 1, 4097, 2048, 33]
    base = synth.make_leaves(2048, seed=4242)
    base_idx = oracle.encode(base, threads=16)
    base_rec = oracle.decode(base_idx, threads=16)
    for step, n in enumerate(sizes):
        if step == 4:
            c.set_chunk_leaves(65536)
            c.set_small_batch_tiles(16)                       # now only <= 512 leaves split (encode), <= 1024 (decode)
        pick = rng.integers(0, 2048, size=n)
        idx = c.encode(base[pick])
        assert np.array_equal(idx, base_idx[pick]), (step, n)
        assert np.array_equal(_bits(c.decode(idx)), _bits(base_rec[pick])), (step, n)
    c.close()


def test_denormal_and_tiny_inputs_match_oracle(codec, oracle):
    """fp32 subnormals are preserved by the MFMA path exactly like the CPU restatement's fmaf chain (no flush-to-zero)."""
    rng = np.random.default_rng(5)
    leaves = np.stack([
        (rng.random(512) * 1e-39).astype(np.float32),                       # all subnormal
        np.where(rng.random(512) < 0.5, 1e-42, 0.0).astype(np.float32),     # sparse subnormal spikes
        (rng.random(512) * 1e-30).astype(np.float32),                       # tiny normals whose products underflow
        np.full(512, np.float32(1.17549435e-38)),                           # smallest normal
    ])
    idx = codec.encode(leaves)
    assert np.array_equal(idx, oracle.encode(leaves, threads=4))
    assert np.array_equal(_bits(codec.decode(idx)), _bits(oracle.decode(idx, threads=4)))


def test_large_path_kernel_variants_agree(pack, oracle, monkeypatch):
    """The large-batch kernels have selectable variants (VQHIP_CONV8 = w8 | rows: 8-wave LDS-plane kernel / row-group kernel for the
    16-channel convs instead of the 16-wave one; VQHIP_STEM=split: gather and GroupNorm as two kernels; VQHIP_STEM=gather: the fused decoder front gathering the (tap, code)
    table through the L1 (stem_fused_k) instead of streaming it through an LDS ring tap by tap (stem_taps_k, the default); VQHIP_CONV4=rows /
    VQHIP_DOWN=rows: the encoder's 4^3 convs on the row kernel — weights in LDS, input rows re-loaded — instead of the LDS plane rings
    conv4_lds_k / conv_down_lds_k; VQHIP_TAIL = slab | groups | rows32: the folded decoder tail skipping structural zeros along depth only
    (conv_mfma32_k), in three plane groups, or with a whole tile per wave, instead of tail_rows16_k).  All of them implement the same arithmetic contract: identical indices, voxels and (debug mode)
    encoder intermediates, and the oracle's on a sample."""
    leaves = np.concatenate([synth.make_leaves(2200, seed=31), synth.sparse_leaves(300, seed=32), synth.edge_leaves()])
    ref_idx = ref_rec = None
    for env in ({}, {"VQHIP_CONV8": "w8"}, {"VQHIP_CONV8": "rows"}, {"VQHIP_STEM": "split"}, {"VQHIP_STEM": "gather"}, {"VQHIP_FIRST": "steps"}, {"VQHIP_FIRST": "roll0"}, {"VQHIP_CONV4": "rows"},
                {"VQHIP_DOWN": "rows"}, {"VQHIP_CONV4": "rows", "VQHIP_DOWN": "rows"}, {"VQHIP_TAIL": "slab"}, {"VQHIP_TAIL": "groups"}, {"VQHIP_TAIL": "rows32"},
                {"VQHIP_FIRST_SRC": "raw"}, {"VQHIP_FIRST_SRC": "raw", "VQHIP_FIRST": "roll0"}):
        for k in ("VQHIP_CONV8", "VQHIP_STEM", "VQHIP_CONV4", "VQHIP_DOWN", "VQHIP_FIRST", "VQHIP_TAIL", "VQHIP_FIRST_SRC"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = HipCodec(pack)
        c.set_small_batch_tiles(0)   # one wave (or workgroup) per tile: the kernels of full chunks, at any batch size
        idx = c.encode(leaves)
        rec = c.decode(idx)
        c.close()
        if ref_idx is None:
            ref_idx, ref_rec = idx, rec
            assert np.array_equal(idx[:256], oracle.encode(leaves[:256], threads=16))
            assert np.array_equal(_bits(rec[:128]), _bits(oracle.decode(idx[:128], threads=16)))
        else:
            assert np.array_equal(idx, ref_idx), env
            assert np.array_equal(_bits(rec), _bits(ref_rec)), env


def test_small_path_kernel_variants_agree(pack, oracle, golden, monkeypatch):
    """Small-batch decode has selectable variants too (VQHIP_TAIL16_TILES=0: folded tail on the 32x32x2 MFMA instead of 16x16x4;
    VQHIP_R64S=stream: 64->64 convs with streamed instead of LDS-resident weights): same voxels bit for bit, and the oracle's."""
    idx = np.concatenate([golden["idx_rand"][:300], golden["idx_edge"]])
    ref = None
    for env in ({}, {"VQHIP_TAIL16_TILES": "0"}, {"VQHIP_R64S": "stream"}):
        for k in ("VQHIP_TAIL16_TILES", "VQHIP_R64S"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = HipCodec(pack)
        rec = c.decode(idx)
        c.close()
        if ref is None:
            ref = rec
            assert np.array_equal(_bits(rec[:64]), _bits(oracle.decode(idx[:64], threads=16)))
        else:
            assert np.array_equal(_bits(rec), _bits(ref)), env


def test_nan_inf_poison_stays_inside_its_own_leaf(pack):
    """SURVEY §8(c) F6 on the GPU (the oracle-level statement is tests/test_oracle_golden.py::test_nan_policy_is_propagation).
    32 leaves share a tile and are the N dimension of every MFMA, GroupNorm partials of a tile sit in one buffer and the VQ
    argmin is a compare/select scan — so: NaN, +Inf, -Inf and 3e38 voxels in leaves 5, 37 (second tile) and 64 (the ragged
    tile's only leaf) of a 65-leaf batch must leave every OTHER leaf's indices and voxels bit-identical to the clean run, on
    both launch paths and through decode.  What the poisoned leaves themselves encode to is reported, not pinned (the reference
    propagates NaN and torch.argmin of a NaN row is unspecified; policy: include/vqvdb_hip.h)."""
    leaves = synth.make_leaves(65, seed=21)
    poisons = {"nan": np.float32(np.nan), "+inf": np.float32(np.inf), "-inf": np.float32(-np.inf), "3e38": np.float32(3e38)}
    bad = [5, 37, 64]
    keep = np.array([i for i in range(65) if i not in bad])
    c = HipCodec(pack)
    try:
        for tiles in (-1, 0):                      # default policy (position-split at this size) / one wave per tile
            c.set_small_batch_tiles(tiles)
            clean_idx = c.encode(leaves)
            clean_rec = c.decode(clean_idx)
            for name, val in poisons.items():
                x = leaves.copy()
                x[5, 100] = val                      # one voxel
                x[37, :] = val                       # a whole leaf
                x[64, 0] = val
                x[64, 511] = -val if name != "nan" else val
                idx = c.encode(x)
                assert np.array_equal(idx[keep], clean_idx[keep]), (tiles, name)
                rec = c.decode(idx)                  # every uint8 is a valid code: decode cannot be poisoned, whatever the leaf encoded to
                assert np.isfinite(rec).all(), (tiles, name)
                assert np.array_equal(_bits(rec[keep]), _bits(clean_rec[keep])), (tiles, name)
                print(f"path {'split' if tiles else 'wave/tile'} poison {name:4s}: leaf 5 -> {np.unique(idx[5]).size} distinct codes, leaf 37 -> codes {np.unique(idx[37])[:4]}, "
                      f"leaf 64 -> {np.unique(idx[64]).size} distinct codes")
    finally:
        c.close()


def test_decompress_the_file_the_reference_writer_wrote(codec):
    """tests/golden/ref_writer_v3.vqvdb comes from the reference's REAL VDBStreamWriter (tools/prove_vqvdb_format.py; 2 grids,
    ragged batches, non-identity transforms): vqhip_decompress_file returns its names / transforms / origins and, per leaf, the
    decode of its indices."""
    import os
    from conftest import ROOT
    from vqvdb_amd import vqvdbfile
    path = os.path.join(ROOT, "tests", "golden", "ref_writer_v3.vqvdb")
    want = vqvdbfile.load(path)
    assert [len(g.origins) for g in want] == [700, 300]
    for batch in (0, 97, 512):
        grids, st = codec.decompress_file(path, batch_leaves=batch)
        assert st["leaves"] == 1000 and st["grids"] == 2
        for (name, tr, org, leaves), g in zip(grids, want):
            assert name == g.name and np.array_equal(tr, g.transform) and np.array_equal(org, g.origins)
            assert np.array_equal(_bits(leaves), _bits(codec.decode(g.indices)))


def test_orchestrator_loop_bench_harness(pack, tmp_path):
    """`leaf_harness loopbench` (bench.py -> orchestrator_loop): the reference orchestrator's serial compress / decompress loops through
    IVQVAECodec::create + the adapter, per-phase timing lines parsed by vqvdb_amd/hostbench.py; the decompress leg checks the leaf count."""
    import os
    import subprocess
    from conftest import ROOT
    from vqvdb_amd import hostbench
    harness = os.path.join(ROOT, "vqvdb_amd", "host", "leaf_harness")
    pk = tmp_path / "w.vqw"
    pk.write_bytes(pack)
    r = subprocess.run([harness, "loopbench", str(pk), "20000", str(tmp_path / "loop.vqvdb"), "64,1000,8192"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    legs = [(m.group(1), int(m.group(2)), int(m.group(3))) for m in hostbench.LOOP_RE.finditer(r.stdout)]
    assert legs == [(d, b, 20000) for b in (64, 1000, 8192) for d in ("compress", "decompress")], r.stdout


def _distinct_batch(n):
    """n DISTINCT leaves: half uniform noise, half background-dominated VDB-like leaves, interleaved so that every tile mixes both."""
    a, b = synth.make_leaves(n // 2, seed=4242), synth.sparse_leaves(n - n // 2, seed=4343)
    x = np.empty((n, 512), dtype=np.float32)
    x[0::2], x[1::2] = a, b
    return x


@pytest.mark.parametrize("regime", ["seed0", "trained"])
def test_full_size_batch_of_distinct_leaves_every_leaf_vs_oracle(regime, oracle, weights):
    """One 65 536-leaf batch in which EVERY tile differs, through the default (one-workgroup-per-tile / persistent) kernels of a full
    chunk — conv8_lds_k's loop across half-tile boundaries, stem_taps_k's ring, the row kernels — compared with the oracle leaf for
    leaf: every index, every voxel, bit for bit.  On the synthetic seed-0 weights and on the checkpoint the reference's own
    optimisation step produced (tests/golden/golden_regimes_v1.npz)."""
    import os
    from conftest import ROOT
    from oracle.oracle import Oracle
    if regime == "seed0":
        w, orc = weights, oracle
    else:
        from test_golden_regimes import regime_weights
        w = regime_weights(np.load(os.path.join(ROOT, "tests", "golden", "golden_regimes_v1.npz")), regime)
        orc = Oracle(w, [t[0] for t in synth.TENSORS])
    n = 65536
    x = _distinct_batch(n)
    assert len(np.unique(x.view(np.dtype((np.void, 2048))).ravel())) > n * 0.9      # distinct leaves (the sparse half holds some exact-zero leaves)
    c = HipCodec(weightpack.dumps(w))
    c.set_chunk_leaves(n)
    threads = len(os.sched_getaffinity(0))
    idx = c.encode(x)
    oidx = orc.encode(x, threads=threads)
    bad = np.nonzero((idx != oidx).any(axis=1))[0]
    assert len(bad) == 0, (regime, len(bad), bad[:8])
    rec = c.decode(idx)
    orec = orc.decode(oidx, threads=threads)
    badv = np.nonzero((_bits(rec) != _bits(orec)).any(axis=1))[0]
    assert len(badv) == 0, (regime, len(badv), badv[:8])
    c.close()


def test_fuzz_slice_launch_paths_agree_bit_for_bit():
    """60 seconds of tools/fuzz_paths.py (random batch sizes 1 ... 70 000 through the default policy, one wave per tile everywhere,
    position-split everywhere; both decoder fronts; adversarial leaves mixed in) — the evidence belongs in the GPU test record,
    not in a log."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_paths.py"), "60"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    assert "fuzz ok" in r.stdout
    rounds = int(r.stdout.split("fuzz ok:")[1].split("rounds")[0])
    assert rounds > 100, r.stdout
    print(r.stdout.strip().splitlines()[-1])


def test_failed_workspace_allocation_leaves_the_handle_usable(pack, oracle):
    """a workspace hipMalloc that fails must (a) fail THAT call with the out-of-memory error, (b) not leave its error in
    the runtime's sticky last-error slot where the next launch check would read it, (c) let the next call succeed.  Device memory is
    hogged through the HIP runtime this process already has loaded (ctypes), not through torch: a second runtime in one process sees no GPU."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    vp, sz = ctypes.c_void_p, ctypes.c_size_t

    def malloc(n):
        p = vp()
        assert hip.hipMalloc(ctypes.byref(p), sz(n)) == 0, n
        return p
    c = HipCodec(pack)
    c.set_chunk_leaves(65536)
    x = synth.make_leaves(64, seed=3)
    want = oracle.encode(x, threads=8)
    assert np.array_equal(c.encode(x), want)              # chunk fitted to the free memory NOW (plenty), small workspace allocated
    big = np.tile(synth.make_leaves(4096, seed=4), (16, 1))
    d_in, d_out = malloc(big.nbytes), malloc(65536 * 64)
    assert hip.hipMemcpy(d_in, vp(big.ctypes.data), sz(big.nbytes), 1) == 0
    free_b, total_b = sz(), sz()
    assert hip.hipMemGetInfo(ctypes.byref(free_b), ctypes.byref(total_b)) == 0
    hog = malloc(free_b.value - (1 << 30))                # leave 1 GiB: the 6.8 GB workspace of a full chunk cannot fit
    with pytest.raises(RuntimeError, match="hipMalloc"):
        c.encode_device(d_in.value, 65536, d_out.value, 0)
    assert hip.hipFree(hog) == 0
    assert np.array_equal(c.encode(x), want)              # the very next call: no stale "out of memory" from the failed allocation
    c.encode_device(d_in.value, 65536, d_out.value, 0)    # and the full chunk fits again
    assert hip.hipDeviceSynchronize() == 0
    got = np.empty((64, 64), dtype=np.uint8)
    assert hip.hipMemcpy(vp(got.ctypes.data), d_out, sz(got.nbytes), 2) == 0
    assert np.array_equal(got, oracle.encode(big[:64], threads=8))
    hip.hipFree(d_in), hip.hipFree(d_out)
    c.close()
