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


# --- C header embedder (role of python/convert_to_header.py:4-44 + src/Bin/bin_model.h) -------------------------------
HEADER_SYMBOL = "g_vqhip_pack_data"      # include/vqvdb_hip_backend.hpp reads g_vqhip_pack_data / g_vqhip_pack_size


def to_header(pack: bytes, source_name: str = "model.vqw") -> str:
    """C/C++ text that embeds a weight pack for ``CodecConfig.source = EmbeddedModel{}``.

    The output is valid as a header (``-DVQVDB_HIP_EMBEDDED_PACK_HEADER='"bin/vqhip_pack.h"'``: the adapter includes it, the way
    TorchBackend.cpp:20 includes bin/bin_model.h) and as a translation unit of its own (``gcc -x c -c vqhip_pack.h`` next to
    ``-DVQVDB_HIP_EMBEDDED_PACK``).  In C++17 the two objects are ``inline const`` (external linkage, one copy however many
    translation units include the header); in C and older C++ they are declared ``extern`` first so that the ``const`` definitions
    keep external linkage; 12 bytes per line like the reference tool.  Define VQHIP_PACK_DECLARE_ONLY to get the declarations alone."""
    loads(pack)                                     # refuses anything that is not a VQWPACK1 pack
    n = len(pack)
    rows = [", ".join(f"0x{b:02x}" for b in pack[i:i + 12]) for i in range(0, n, 12)]
    return (
        "#ifndef VQHIP_PACK_H_INCLUDED\n#define VQHIP_PACK_H_INCLUDED\n\n#include <stddef.h>\n\n"
        f"/* Weight pack: {source_name}\n * Size:        {n} bytes (VQWPACK1, vqvdb_amd/weightpack.py) */\n\n"
        "#if defined(__cplusplus) && __cplusplus >= 201703L\n"
        "#define VQHIP_PACK_OBJECT inline const /* C++17 inline variable: external linkage, ONE copy however many TUs include this */\n"
        "#else\n#define VQHIP_PACK_OBJECT const\n#endif\n"
        "#ifdef __cplusplus\nextern \"C\" {\n#endif\n"
        "#ifdef VQHIP_PACK_DECLARE_ONLY\n"
        f"extern const unsigned char {HEADER_SYMBOL}[{n}];\nextern const size_t g_vqhip_pack_size;\n"
        "#else\n"
        "#if !(defined(__cplusplus) && __cplusplus >= 201703L)\n"
        f"extern const unsigned char {HEADER_SYMBOL}[{n}]; /* declared extern first: the const definitions keep external linkage */\n"
        "extern const size_t g_vqhip_pack_size;\n#endif\n"
        f"VQHIP_PACK_OBJECT size_t g_vqhip_pack_size = {n};\n"
        f"VQHIP_PACK_OBJECT unsigned char {HEADER_SYMBOL}[{n}] = {{\n    " + ",\n    ".join(rows) + "\n};\n"
        "#endif\n#ifdef __cplusplus\n}\n#endif\n#undef VQHIP_PACK_OBJECT\n\n#endif /* VQHIP_PACK_H_INCLUDED */\n")


def save_header(pack_path, header_path) -> int:
    import os
    with open(pack_path, "rb") as f:
        pack = f.read()
    with open(header_path, "w") as f:
        f.write(to_header(pack, os.path.basename(str(pack_path))))
    return len(pack)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="VQWPACK1 tools: --header embeds a weight pack into a C/C++ header for EmbeddedModel builds")
    ap.add_argument("--header", nargs=2, metavar=("MODEL.vqw", "OUT.h"), required=True)
    a = ap.parse_args()
    print(f"{a.header[1]}: {save_header(*a.header)} bytes as {HEADER_SYMBOL}[] / g_vqhip_pack_size")
