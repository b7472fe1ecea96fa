"""`.vqvdb` v3 container in numpy (byte layout: SURVEY.md App. B; reference writer/reader
src/Utils/VQVDB_Reader.cpp:81-150,168-300).  Host-side framing only — no codec work happens here.

    file : "VQVDB" | u8 version=3 | u8 numGrids | u32 numEmbeddings | u8 latentDimCount
    grid : u32 nameLength | name | f32 transform[16] | u16 latentShape[latentDimCount] | u32 totalBlocks
           totalBlocks x { i32 origin[3] | u8 indices[prod(latentShape)] }        (76 B per leaf)
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List

import numpy as np

MAGIC = b"VQVDB"
VERSION = 3
RECORD = np.dtype([("origin", "<i4", (3,)), ("indices", "u1", (64,))])
assert RECORD.itemsize == 76

IDENTITY = np.eye(4, dtype=np.float32).reshape(16)


@dataclass
class Grid:
    name: str
    origins: np.ndarray                      # int32 [n,3]
    indices: np.ndarray                      # uint8 [n,64]
    transform: np.ndarray = field(default_factory=lambda: IDENTITY.copy())
    latent_shape: tuple = (4, 4, 4)


def dumps(grids: List[Grid], num_embeddings: int = 256) -> bytes:
    if not 1 <= len(grids) <= 255:
        raise ValueError("a .vqvdb file holds 1..255 grids")
    out = [MAGIC + struct.pack("<BBIB", VERSION, len(grids), num_embeddings, 3)]
    for g in grids:
        n = len(g.origins)
        if n >= 1 << 32:
            raise ValueError("more than 2^32-1 leaves in one grid")
        name = g.name.encode()
        out.append(struct.pack("<I", len(name)) + name)
        out.append(np.asarray(g.transform, dtype="<f4").reshape(16).tobytes())
        out.append(struct.pack("<3H", *g.latent_shape) + struct.pack("<I", n))
        rec = np.empty(n, dtype=RECORD)
        rec["origin"] = np.asarray(g.origins, dtype=np.int32).reshape(n, 3)
        rec["indices"] = np.asarray(g.indices, dtype=np.uint8).reshape(n, 64)
        out.append(rec.tobytes())
    return b"".join(out)


def loads(buf: bytes) -> List[Grid]:
    if len(buf) < 12:
        raise ValueError("Failed to read file header.")
    if buf[:5] != MAGIC:
        raise ValueError("Invalid file magic; not a .vqvdb file.")
    version, n_grids, _num_emb, dim_count = struct.unpack_from("<BBIB", buf, 5)
    if version != VERSION:
        raise ValueError(f"Unsupported .vqvdb version {version} (expected 3).")
    off, grids = 12, []
    for _ in range(n_grids):
        (name_len,) = struct.unpack_from("<I", buf, off); off += 4
        name = buf[off:off + name_len].decode(); off += name_len
        transform = np.frombuffer(buf, dtype="<f4", count=16, offset=off).copy(); off += 64
        shape = struct.unpack_from(f"<{dim_count}H", buf, off); off += 2 * dim_count
        (total,) = struct.unpack_from("<I", buf, off); off += 4
        block = int(np.prod(shape))
        if block != 64:
            raise ValueError(f"grid '{name}' has latent shape {list(shape)}; only [4,4,4] is supported")
        if off + total * 76 > len(buf):
            raise ValueError("File truncated: incomplete block data.")
        rec = np.frombuffer(buf, dtype=RECORD, count=total, offset=off); off += total * 76
        grids.append(Grid(name, rec["origin"].copy(), rec["indices"].copy(), transform, tuple(shape)))
    return grids


def save(path, grids: List[Grid], num_embeddings: int = 256) -> None:
    with open(path, "wb") as f:
        f.write(dumps(grids, num_embeddings))


def load(path) -> List[Grid]:
    with open(path, "rb") as f:
        return loads(f.read())
