"""CPU-only checks of the host side: weight pack round trip, the C-ABI library loads and
exports every symbol include/vqvdb_hip.h declares, header/ctypes agreement, and the
no-GPU failure mode (loud error, no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from vqvdb_amd import codec as vc
from vqvdb_amd import synth, weightpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_weightpack_roundtrip(weights, tmp_path):
    blob = weightpack.dumps(weights)
    back = weightpack.loads(blob)
    assert list(back) == [t[0] for t in synth.TENSORS]
    for k, v in weights.items():
        assert back[k].shape == v.shape and np.array_equal(back[k], v)
    p = tmp_path / "model.vqw"
    weightpack.save(p, weights)
    assert np.array_equal(weightpack.load(p)["encoder.down.weight"], weights["encoder.down.weight"])
    with pytest.raises(ValueError):
        weightpack.loads(b"garbage-garbage-")


def test_from_state_dict_drops_training_buffers(weights):
    sd = dict(weights)
    sd["quantizer.cluster_size"] = np.ones(256, np.float32)
    sd["quantizer.embed_avg"] = weights["quantizer.embedding"].copy()
    out = weightpack.from_state_dict(sd)
    assert set(out) == set(weights)


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "vqvdb_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vqhip_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from vqvdb_amd.build import build
    lib = ctypes.CDLL(build())
    declared = _declared_symbols()
    assert sorted(vc.ABI_SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in ctypes.cast(lib.vqhip_version, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_create_fails_loudly_without_gpu_or_pack(weights, capsys):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the failure path is covered by the -m gpu tests")
    with pytest.raises(RuntimeError, match="no HIP device|hip"):
        vc.HipCodec(weightpack.dumps(weights))
    be = vc.IVQVAECodec.create(vc.CodecConfig(device=vc.CodecConfig.Device.CUDA, source=weightpack.dumps(weights)), vc.BackendType.HIP)
    assert be is None and "Failed to create VQ-VAE backend" in capsys.readouterr().err
    with pytest.raises(RuntimeError, match="bad magic"):
        vc.HipCodec(b"x" * 64)


def test_vqvdb_stream_framing_roundtrip(tmp_path):
    """The OpenVDB-free .vqvdb v3 reader/writer (vqvdb_amd/host/vqvdb_stream.hpp) against the byte
    layout of SURVEY.md App. B, parsed independently here."""
    import struct
    import subprocess
    from vqvdb_amd.build import build, build_harness
    build()
    exe = build_harness()
    path = tmp_path / "t.vqvdb"
    assert subprocess.run([exe, "streamtest", str(path)], capture_output=True).returncode == 0
    raw = path.read_bytes()
    magic, ver, ngrids, nemb, ndim = struct.unpack_from("<5sBBIB", raw, 0)
    assert (magic, ver, ngrids, nemb, ndim) == (b"VQVDB", 3, 2, 256, 3)
    off, total = 12, 0
    for expect_name, expect_blocks in ((b"density", 700), (b"temperature", 300)):
        (nl,) = struct.unpack_from("<I", raw, off); off += 4
        assert raw[off:off + nl] == expect_name; off += nl
        off += 64
        assert struct.unpack_from("<3H", raw, off) == (4, 4, 4); off += 6
        (nb,) = struct.unpack_from("<I", raw, off); off += 4
        assert nb == expect_blocks
        x, y, z = struct.unpack_from("<3i", raw, off)           # first chunk: origin then 64 index bytes
        assert (x, y, z) == (8 * (total % 1024), 8 * ((total // 1024) % 1024), 0)
        off += nb * 76; total += nb
    assert off == len(raw) and total == 1000


def test_vqvdbfile_numpy_matches_cpp_writer(tmp_path):
    """vqvdb_amd/vqvdbfile.py (numpy framing used by the stream parity tests) parses the file the C++ writer
    produced and re-serialises it to the same bytes; truncated / foreign files are rejected with the reader's messages."""
    import subprocess
    from vqvdb_amd import vqvdbfile
    from vqvdb_amd.build import build, build_harness
    build()
    exe = build_harness()
    path = tmp_path / "t.vqvdb"
    assert subprocess.run([exe, "streamtest", str(path)], capture_output=True).returncode == 0
    raw = path.read_bytes()
    grids = vqvdbfile.loads(raw)
    assert [g.name for g in grids] == ["density", "temperature"] and [len(g.origins) for g in grids] == [700, 300]
    assert grids[1].transform[0] == 1.5 and grids[0].indices.dtype == np.uint8
    assert vqvdbfile.dumps(grids) == raw
    with pytest.raises(ValueError, match="truncated"):
        vqvdbfile.loads(raw[:-10])
    with pytest.raises(ValueError, match="magic"):
        vqvdbfile.loads(b"OPENVDB" + raw[7:])
    empty = [vqvdbfile.Grid("e", np.zeros((0, 3), np.int32), np.zeros((0, 64), np.uint8))]
    assert len(vqvdbfile.loads(vqvdbfile.dumps(empty))[0].origins) == 0


def test_full_training_host_helpers(weights):
    """Flat parameter vector <-> state_dict (the reference's parameter order) and the closed-form cosine schedule."""
    import math
    import torch
    from vqvdb_amd import full_training as ft
    flat = ft.dict_to_flat(weights)
    assert flat.size == 995905 and flat.dtype == np.float32
    back = ft.flat_to_dict(flat)
    assert list(back) == [n for n, _ in ft.TRAINABLE] and all(np.array_equal(back[k], weights[k]) for k in back)
    assert "quantizer.embedding" not in back
    with pytest.raises(ValueError):
        ft.flat_to_dict(flat[:-1])
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=50)
    for step in range(50):
        assert math.isclose(ft.cosine_lr(1e-4, step, 50), sch.get_last_lr()[0], rel_tol=1e-9, abs_tol=1e-15)
        opt.step()
        sch.step()
