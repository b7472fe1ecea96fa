"""Raw-fp32 weight pack ("VQWPACK1") — the model artefact the HIP backend loads.

Plays the role of the reference's TorchScript / ONNX blobs
(python/save_for_inference.py:114-140, python/to_onnx.py:59-182,
src/Bin/bin_model.h) without any ML runtime at inference: a flat table of named
fp32 tensors.  ``CodecConfig.source = std::filesystem::path`` points at such a
file (include/vqvdb_hip.h: ``vqhip_create``).

Layout (little-endian):
    char     magic[8]  = "VQWPACK1"
    uint32   n_tensors
    uint32   reserved  = 0
    entry[n] : char name[64] (NUL padded) | uint32 ndim | uint32 dims[6]
               | uint64 offset (bytes from file start, 64-B aligned) | uint64 count
    data     : fp32
"""
from __future__ import annotations

import struct
from typing import Mapping

import numpy as np

MAGIC = b"VQWPACK1"
_ENTRY = struct.Struct("<64sI6IQQ")   # 64 + 4 + 24 + 8 + 8 = 108 bytes
_HEAD = struct.Struct("<8sII")


def dumps(tensors: Mapping[str, np.ndarray]) -> bytes:
    names = list(tensors)
    table_end = _HEAD.size + _ENTRY.size * len(names)
    off = (table_end + 63) // 64 * 64
    entries, blobs = [], []
    for name in names:
        a = np.ascontiguousarray(np.asarray(tensors[name], dtype="<f4"))
        if a.ndim > 6 or len(name.encode()) > 63:
            raise ValueError(f"tensor {name!r}: rank/name too large for the pack")
        dims = list(a.shape) + [0] * (6 - a.ndim)
        entries.append(_ENTRY.pack(name.encode(), a.ndim, *dims, off, a.size))
        blobs.append((off, a.tobytes()))
        off = (off + a.nbytes + 63) // 64 * 64
    buf = bytearray(off)
    buf[:_HEAD.size] = _HEAD.pack(MAGIC, len(names), 0)
    pos = _HEAD.size
    for e in entries:
        buf[pos:pos + _ENTRY.size] = e
        pos += _ENTRY.size
    for o, b in blobs:
        buf[o:o + len(b)] = b
    return bytes(buf)


def loads(data: bytes) -> dict[str, np.ndarray]:
    magic, n, _ = _HEAD.unpack_from(data, 0)
    if magic != MAGIC:
        raise ValueError("not a VQWPACK1 weight pack")
    out = {}
    for i in range(n):
        rec = _ENTRY.unpack_from(data, _HEAD.size + i * _ENTRY.size)
        name = rec[0].rstrip(b"\0").decode()
        ndim, dims, off, count = rec[1], rec[2:8], rec[8], rec[9]
        a = np.frombuffer(data, dtype="<f4", count=count, offset=off)
        out[name] = a.reshape(dims[:ndim]).copy()
    return out


def save(path, tensors: Mapping[str, np.ndarray]) -> None:
    with open(path, "wb") as f:
        f.write(dumps(tensors))


def load(path) -> dict[str, np.ndarray]:
    with open(path, "rb") as f:
        return loads(f.read())


def from_state_dict(state_dict) -> dict[str, np.ndarray]:
    """Export a trained reference ``VQVAE.state_dict()`` (torch tensors or arrays)
    to pack tensors.  Training-only buffers (``quantizer.cluster_size``,
    ``quantizer.embed_avg``; VQVAE_v2.py:103-105) are dropped."""
    out = {}
    for k, v in state_dict.items():
        if k in ("quantizer.cluster_size", "quantizer.embed_avg"):
            continue
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        out[k] = a.astype(np.float32)
    return out
