"""Host-side mirror of the reference's codec interface over the C ABI of libvqvdb_hip.so.

Mirrors src/core/IVQVAECodec.hpp (``BackendType``, ``DataType``, ``TensorView``, ``Tensor``,
``CodecConfig``, ``IVQVAECodec.create/encode/decode/getLatentShape``) with the same names,
argument meaning and error behaviour, so the parity tests read like the reference's call
sites (src/orchestrator/VQVAECodec.cpp:108-127,166-196).  The C++ adapter a maintainer would
compile into the reference tree is include/vqvdb_hip_backend.hpp; this module is the same
thing for Python callers, tests and bench.py.

There is NO CPU fallback: if the shared library or a gfx950 device is missing, ``create``
reports the failure and returns ``None`` exactly like the reference factory
(src/core/IVQVAECodec.cpp:106-109), and ``HipCodec`` raises.
"""
from __future__ import annotations

import ctypes
import enum
import os
import sys
from dataclasses import dataclass, field
from typing import Optional, Sequence, Union

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvqvdb_hip.so")

LEAF_VOXELS = 512
LATENT_VOXELS = 64


class BackendType(enum.Enum):   # IVQVAECodec.hpp:21, HIP appended (existing values unchanged)
    LibTorch = 0
    ONNX = 1
    HIP = 2


class DataType(enum.Enum):      # IVQVAECodec.hpp:38-41
    FLOAT32 = 0
    UINT8 = 1


class EmbeddedModel:            # IVQVAECodec.hpp:27
    pass


EMBEDDED_PACK_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "embedded_model.vqw")


@dataclass
class TensorView:               # IVQVAECodec.hpp:49-53 — non-owning view of host data
    data: np.ndarray
    shape: Sequence[int]
    dtype: DataType


@dataclass
class Tensor:                   # IVQVAECodec.hpp:61-80 — owning result
    buffer: np.ndarray          # flat uint8 bytes
    shape: list
    dtype: DataType

    def getData(self) -> np.ndarray:
        t = np.float32 if self.dtype == DataType.FLOAT32 else np.uint8
        return self.buffer.view(t).reshape(self.shape)


@dataclass
class CodecConfig:              # IVQVAECodec.hpp:85-89
    class Device(enum.Enum):
        CPU = 0
        CUDA = 1                # read as "GPU": the HIP backend only accepts this value

    device: "CodecConfig.Device" = None
    source: Union[EmbeddedModel, str, os.PathLike, bytes] = field(default_factory=EmbeddedModel)
    device_id: int = 0          # extension: HIP device ordinal (reference hard-codes 0, OnnxBackend_Cuda.cpp:21)

    def __post_init__(self):
        if self.device is None:
            self.device = CodecConfig.Device.CPU


class _KernelStat(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("launches", ctypes.c_int64), ("total_ms", ctypes.c_double),
                ("flops_per_leaf", ctypes.c_double), ("eff_flops_per_leaf", ctypes.c_double), ("leaves", ctypes.c_int64)]


# every symbol include/vqvdb_hip.h declares
ABI_SYMBOLS = [
    "vqhip_create", "vqhip_destroy", "vqhip_last_error", "vqhip_latent_shape", "vqhip_encode", "vqhip_decode",
    "vqhip_encode_device", "vqhip_decode_device", "vqhip_encode_leaves", "vqhip_decode_leaves", "vqhip_set_chunk_leaves", "vqhip_profile_enable",
    "vqhip_profile_read", "vqhip_debug_enable", "vqhip_debug_fetch", "vqhip_selftest_mfma", "vqhip_version",
    "vqhip_multi_create", "vqhip_multi_destroy", "vqhip_multi_last_error", "vqhip_multi_encode", "vqhip_multi_decode",
    "vqhip_decompress_file", "vqhip_compress_file", "vqhip_reserve",
    "vqhip_train_begin", "vqhip_train_vq_stats_device", "vqhip_train_vq_update_device", "vqhip_train_get_state", "vqhip_train_set_state",
    "vqhip_train_commit", "vqhip_set_small_batch_tiles", "vqhip_train_eval_device",
    "vqhip_fulltrain_begin", "vqhip_fulltrain_param_count", "vqhip_fulltrain_forward_device", "vqhip_fulltrain_fwdbwd_device",
    "vqhip_fulltrain_apply_device", "vqhip_fulltrain_get_params", "vqhip_fulltrain_set_params",
    "vqhip_fulltrain_get_opt_state", "vqhip_fulltrain_set_opt_state", "vqhip_workspace_bytes", "vqhip_chunk_leaves", "vqhip_multi_worker_info", "vqhip_fulltrain_fwdbwd_overlap_device", "vqhip_fulltrain_decoder_offset", "vqhip_fulltrain_ready_stream", "vqhip_fulltrain_set_folded_tail",
]

class _GridInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("transform", ctypes.c_float * 16), ("latent_shape", ctypes.c_int64 * 3),
                ("num_embeddings", ctypes.c_uint32), ("total_blocks", ctypes.c_uint64), ("grid_index", ctypes.c_int)]


class StreamStats(ctypes.Structure):
    """vqhip_stream_stats: where a whole-file compress/decompress spent its time."""
    _fields_ = [("leaves", ctypes.c_int64), ("grids", ctypes.c_int32), ("wall_s", ctypes.c_double), ("read_s", ctypes.c_double),
                ("alloc_s", ctypes.c_double), ("copy_s", ctypes.c_double), ("io_wait_s", ctypes.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class _GridSource(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("transform", ctypes.POINTER(ctypes.c_float)), ("leaf_ptrs", ctypes.c_void_p),
                ("origins", ctypes.c_void_p), ("n_leaves", ctypes.c_int64)]


GRID_BEGIN_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(_GridInfo))
LEAF_ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_int64,
                                 ctypes.POINTER(ctypes.c_void_p))

PHASE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)

_lib = None


def load_library() -> ctypes.CDLL:
    """dlopen libvqvdb_hip.so (built in-tree by ``python -m vqvdb_amd.build``) and type its ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m vqvdb_amd.build` (no CPU fallback exists)")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    lib.vqhip_create.argtypes = [ctypes.c_char_p, vp, ctypes.c_size_t, ci, ctypes.POINTER(vp)]
    lib.vqhip_destroy.argtypes = [vp]
    lib.vqhip_destroy.restype = None
    lib.vqhip_last_error.argtypes = [vp]
    lib.vqhip_last_error.restype = ctypes.c_char_p
    lib.vqhip_latent_shape.argtypes = [vp, ctypes.POINTER(i64)]
    lib.vqhip_encode.argtypes = [vp, vp, i64, vp]
    lib.vqhip_decode.argtypes = [vp, vp, i64, vp]
    lib.vqhip_encode_device.argtypes = [vp, vp, i64, vp, vp]
    lib.vqhip_decode_device.argtypes = [vp, vp, i64, vp, vp]
    lib.vqhip_encode_leaves.argtypes = [vp, vp, i64, vp]
    lib.vqhip_decode_leaves.argtypes = [vp, vp, i64, vp]
    lib.vqhip_set_chunk_leaves.argtypes = [vp, i64]
    lib.vqhip_reserve.argtypes = [vp, i64]
    lib.vqhip_set_small_batch_tiles.argtypes = [vp, ci]
    lib.vqhip_train_begin.argtypes = [vp, vp, vp]
    lib.vqhip_train_vq_stats_device.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    lib.vqhip_train_vq_update_device.argtypes = [vp, vp, ctypes.c_float, ctypes.c_float, vp]
    lib.vqhip_train_eval_device.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    lib.vqhip_train_get_state.argtypes = [vp, vp, vp, vp]
    lib.vqhip_train_set_state.argtypes = [vp, vp, vp, vp]
    lib.vqhip_train_commit.argtypes = [vp]
    lib.vqhip_fulltrain_begin.argtypes = [vp]
    lib.vqhip_fulltrain_param_count.argtypes = [vp]
    lib.vqhip_fulltrain_param_count.restype = i64
    lib.vqhip_fulltrain_forward_device.argtypes = [vp, vp, i64, vp]
    lib.vqhip_fulltrain_fwdbwd_device.argtypes = [vp, vp, i64, i64, vp, vp, vp]
    cf = ctypes.c_float
    lib.vqhip_fulltrain_apply_device.argtypes = [vp, vp, vp, cf, i64, cf, cf, cf, cf, cf, cf, vp]
    lib.vqhip_fulltrain_get_params.argtypes = [vp, vp]
    lib.vqhip_fulltrain_set_params.argtypes = [vp, vp]
    lib.vqhip_multi_worker_info.argtypes = [vp, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci)]
    lib.vqhip_fulltrain_fwdbwd_overlap_device.argtypes = [vp, vp, i64, i64, vp, vp, vp, PHASE_FN, vp]
    lib.vqhip_fulltrain_set_folded_tail.argtypes = [vp, ci]
    lib.vqhip_fulltrain_decoder_offset.argtypes = [vp]
    lib.vqhip_fulltrain_decoder_offset.restype = ctypes.c_int64
    lib.vqhip_fulltrain_ready_stream.argtypes = [vp]
    lib.vqhip_fulltrain_ready_stream.restype = ctypes.c_void_p
    lib.vqhip_workspace_bytes.argtypes = [vp]
    lib.vqhip_workspace_bytes.restype = ctypes.c_int64
    lib.vqhip_chunk_leaves.argtypes = [vp]
    lib.vqhip_chunk_leaves.restype = ctypes.c_int64
    lib.vqhip_fulltrain_get_opt_state.argtypes = [vp, vp, vp]
    lib.vqhip_fulltrain_set_opt_state.argtypes = [vp, vp, vp]
    lib.vqhip_profile_enable.argtypes = [vp, ci]
    lib.vqhip_profile_read.argtypes = [vp, ctypes.POINTER(_KernelStat), ci, ctypes.POINTER(ci)]
    lib.vqhip_debug_enable.argtypes = [vp, ci]
    lib.vqhip_debug_fetch.argtypes = [vp, ctypes.c_char_p, i64, vp]
    lib.vqhip_selftest_mfma.argtypes = [vp, ctypes.POINTER(i64)]
    lib.vqhip_version.restype = ctypes.c_char_p
    lib.vqhip_multi_create.argtypes = [ctypes.c_char_p, vp, ctypes.c_size_t, ctypes.POINTER(ci), ci, ctypes.POINTER(vp)]
    lib.vqhip_multi_destroy.argtypes = [vp]
    lib.vqhip_multi_destroy.restype = None
    lib.vqhip_multi_last_error.argtypes = [vp]
    lib.vqhip_multi_last_error.restype = ctypes.c_char_p
    lib.vqhip_multi_encode.argtypes = [vp, vp, i64, vp]
    lib.vqhip_multi_decode.argtypes = [vp, vp, i64, vp]
    lib.vqhip_decompress_file.argtypes = [vp, ctypes.c_char_p, i64, GRID_BEGIN_FN, LEAF_ALLOC_FN, vp, ctypes.POINTER(StreamStats)]
    lib.vqhip_compress_file.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(_GridSource), ci, i64, ctypes.POINTER(StreamStats)]
    for name in ABI_SYMBOLS:
        if getattr(lib, name).argtypes is None and name not in ("vqhip_version",):
            raise RuntimeError(f"codec.py: no argtypes declared for {name} (pointers would be truncated to 32 bits)")
        if name not in ("vqhip_destroy", "vqhip_last_error", "vqhip_version", "vqhip_multi_destroy", "vqhip_multi_last_error",
                        "vqhip_fulltrain_param_count", "vqhip_workspace_bytes", "vqhip_chunk_leaves", "vqhip_fulltrain_decoder_offset",
                        "vqhip_fulltrain_ready_stream"):
            getattr(lib, name).restype = ci
    _lib = lib
    return lib


class HipCodec:
    """Thin owner of a ``vqhip_codec*`` — the C ABI one-to-one, numpy/raw pointers in and out."""

    def __init__(self, pack: Union[str, os.PathLike, bytes], device_id: int = 0):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        if isinstance(pack, (bytes, bytearray, memoryview)):
            self._pack = bytes(pack)
            rc = self._lib.vqhip_create(None, self._pack, len(self._pack), device_id, ctypes.byref(self._h))
        else:
            rc = self._lib.vqhip_create(os.fspath(pack).encode(), None, 0, device_id, ctypes.byref(self._h))
        if rc != 0:
            raise RuntimeError(self._lib.vqhip_last_error(None).decode())

    def _check(self, rc: int):
        if rc != 0:
            raise RuntimeError(self._lib.vqhip_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.vqhip_destroy(self._h)
            self._h = ctypes.c_void_p()

    __del__ = close

    def latent_shape(self) -> list:
        out = (ctypes.c_int64 * 3)()
        self._check(self._lib.vqhip_latent_shape(self._h, out))
        return list(out)

    @staticmethod
    def _out(out, shape, dtype):
        if out is None:
            return np.empty(shape, dtype=dtype)
        if out.dtype != dtype or out.size != shape[0] * shape[1] or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous {np.dtype(dtype).name} array of {shape[0] * shape[1]} elements")
        return out

    def encode(self, leaves: np.ndarray, out: Optional[np.ndarray] = None) -> np.ndarray:
        leaves = np.ascontiguousarray(leaves, dtype=np.float32).reshape(-1, LEAF_VOXELS)
        idx = self._out(out, (leaves.shape[0], LATENT_VOXELS), np.uint8)
        self._check(self._lib.vqhip_encode(self._h, leaves.ctypes.data, leaves.shape[0], idx.ctypes.data))
        return idx

    def decode(self, indices: np.ndarray, out: Optional[np.ndarray] = None) -> np.ndarray:
        indices = np.ascontiguousarray(indices, dtype=np.uint8).reshape(-1, LATENT_VOXELS)
        out = self._out(out, (indices.shape[0], LEAF_VOXELS), np.float32)
        self._check(self._lib.vqhip_decode(self._h, indices.ctypes.data, indices.shape[0], out.ctypes.data))
        return out

    @staticmethod
    def _leaf_ptrs(leaf_arrays: Sequence[np.ndarray], n: int, writable: bool):
        """Addresses of n per-leaf buffers, each checked: float32, 512 elements, C-contiguous (the library reads / writes 2 KiB each)."""
        if len(leaf_arrays) != n:
            raise ValueError(f"expected {n} leaf buffers, got {len(leaf_arrays)}")
        for i, a in enumerate(leaf_arrays):
            if not isinstance(a, np.ndarray) or a.dtype != np.float32 or a.size != LEAF_VOXELS or not a.flags.c_contiguous \
                    or (writable and not a.flags.writeable):
                raise ValueError(f"leaf buffer {i}: need a C-contiguous{' writable' if writable else ''} float32 array of {LEAF_VOXELS} elements")
        return (ctypes.c_void_p * n)(*[a.ctypes.data for a in leaf_arrays])

    def encode_leaves(self, leaf_arrays: Sequence[np.ndarray]) -> np.ndarray:
        """Leaf-pointer entry point: one 512-float buffer per leaf (e.g. OpenVDB leaf buffers)."""
        n = len(leaf_arrays)
        ptrs = self._leaf_ptrs(leaf_arrays, n, writable=False)
        idx = np.empty((n, LATENT_VOXELS), dtype=np.uint8)
        self._check(self._lib.vqhip_encode_leaves(self._h, ptrs, n, idx.ctypes.data))
        return idx

    def decode_leaves(self, indices: np.ndarray, leaf_arrays: Sequence[np.ndarray]) -> None:
        indices = np.ascontiguousarray(indices, dtype=np.uint8).reshape(-1, LATENT_VOXELS)
        n = indices.shape[0]
        ptrs = self._leaf_ptrs(leaf_arrays, n, writable=True)
        self._check(self._lib.vqhip_decode_leaves(self._h, indices.ctypes.data, n, ptrs))

    def compress_file(self, path, grids, batch_leaves: int = 0) -> dict:
        """Whole-file compress (vqhip_compress_file).  grids: sequence of (name, origins int32 [n,3], leaves float32 [n,512]
        or a list of n 512-float arrays, transform or None).  Returns the stream statistics."""
        n_g = len(grids)
        src = (_GridSource * max(n_g, 1))()
        keep = []
        for i, (name, origins, leaves, transform) in enumerate(grids):
            origins = np.ascontiguousarray(origins, dtype=np.int32).reshape(-1, 3)
            n = origins.shape[0]
            if isinstance(leaves, np.ndarray):
                leaves = np.ascontiguousarray(leaves, dtype=np.float32).reshape(n, LEAF_VOXELS)
                ptrs = leaves.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(LEAF_VOXELS * 4)
            else:
                if len(leaves) != n:
                    raise ValueError("one leaf buffer per origin")
                self._leaf_ptrs(leaves, n, writable=False)   # validates every buffer
                ptrs = np.array([a.ctypes.data for a in leaves], dtype=np.uint64)
            ptrs = np.ascontiguousarray(ptrs, dtype=np.uint64)
            tr = None if transform is None else np.ascontiguousarray(transform, dtype=np.float32).reshape(16)
            bname = name.encode()
            keep += [origins, leaves, ptrs, tr, bname]
            src[i].name = bname
            src[i].transform = tr.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if tr is not None else None
            src[i].leaf_ptrs = ptrs.ctypes.data if n else None
            src[i].origins = origins.ctypes.data if n else None
            src[i].n_leaves = n
        st = StreamStats()
        self._check(self._lib.vqhip_compress_file(self._h, os.fspath(path).encode(), src, n_g, batch_leaves, ctypes.byref(st)))
        return st.as_dict()

    def decompress_file(self, path, batch_leaves: int = 0, out: Optional[np.ndarray] = None):
        """Whole-file decompress (vqhip_decompress_file).  Returns ([(name, transform[16], origins [n,3], leaves [n,512])], stats);
        the leaf allocator hands out one fresh 2 KiB-per-leaf block per batch, the stand-in for tree.touchLeaf().
        out: optional preallocated C-contiguous float32 [total_leaves, 512] pool — leaves of all grids are then placed in it in
        file order (no per-batch allocation, no concatenation: multi-million-leaf files) and the returned leaf arrays are views."""
        grids, blocks = [], []
        if out is not None and (out.dtype != np.float32 or out.ndim != 2 or out.shape[1] != LEAF_VOXELS or not out.flags.c_contiguous):
            raise ValueError("out must be a C-contiguous float32 [n, 512] array")
        cursor = [0]

        def on_grid(_user, gi):
            g = gi.contents
            grids.append([g.name.decode(), np.array(g.transform[:], dtype=np.float32), int(g.total_blocks)])
            blocks.append([])
            return 0

        def on_alloc(_user, gidx, origins, n, out_ptrs):
            try:
                org = np.ctypeslib.as_array(origins, shape=(n, 3)).copy()
                if out is not None:
                    if cursor[0] + n > out.shape[0]:
                        raise ValueError(f"out holds {out.shape[0]} leaves, the file has more")
                    buf = out[cursor[0]:cursor[0] + n]
                    cursor[0] += n
                else:
                    buf = np.empty((n, LEAF_VOXELS), dtype=np.float32)
                dst = np.ctypeslib.as_array(ctypes.cast(out_ptrs, ctypes.POINTER(ctypes.c_uint64)), shape=(n,))
                dst[:] = buf.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(LEAF_VOXELS * 4)
                blocks[gidx].append((org, buf))
                return 0
            except Exception as e:  # noqa: BLE001 — an exception must not unwind through the C frames
                print(f"decompress_file allocator: {e}", file=sys.stderr)
                return 1

        cb_g, cb_a = GRID_BEGIN_FN(on_grid), LEAF_ALLOC_FN(on_alloc)
        st = StreamStats()
        self._check(self._lib.vqhip_decompress_file(self._h, os.fspath(path).encode(), batch_leaves, cb_g, cb_a, None, ctypes.byref(st)))
        result = []
        for (name, tr, _total), bl in zip(grids, blocks):
            org = np.concatenate([b[0] for b in bl]) if bl else np.zeros((0, 3), np.int32)
            if out is not None and len(bl):   # consecutive slices of the pool: one view, no copy
                first = (bl[0][1].ctypes.data - out.ctypes.data) // (LEAF_VOXELS * 4)
                lv = out[first:first + sum(len(b[1]) for b in bl)]
            else:
                lv = np.concatenate([b[1] for b in bl]) if bl else np.zeros((0, LEAF_VOXELS), np.float32)
            result.append((name, tr, org, lv))
        return result, st.as_dict()

    def encode_device(self, leaves_ptr: int, n: int, idx_ptr: int, stream: int = 0):
        self._check(self._lib.vqhip_encode_device(self._h, leaves_ptr, n, idx_ptr, stream or None))

    def decode_device(self, idx_ptr: int, n: int, leaves_ptr: int, stream: int = 0):
        self._check(self._lib.vqhip_decode_device(self._h, idx_ptr, n, leaves_ptr, stream or None))

    def set_chunk_leaves(self, n: int):
        self._check(self._lib.vqhip_set_chunk_leaves(self._h, n))

    # ---- codebook training (VectorQuantizerEMA in training mode; see vqvdb_amd/codebook_training.py) ----
    def train_begin(self, cluster_size: Optional[np.ndarray] = None, embed_avg: Optional[np.ndarray] = None):
        cs = None if cluster_size is None else np.ascontiguousarray(cluster_size, dtype=np.float32).reshape(256)
        av = None if embed_avg is None else np.ascontiguousarray(embed_avg, dtype=np.float32).reshape(256, 128)
        self._check(self._lib.vqhip_train_begin(self._h, None if cs is None else cs.ctypes.data, None if av is None else av.ctypes.data))

    def train_vq_stats_device(self, leaves_ptr: int, n: int, stats_ptr: int, idx_ptr: int = 0, latent_ptr: int = 0, stream: int = 0):
        self._check(self._lib.vqhip_train_vq_stats_device(self._h, leaves_ptr, n, stats_ptr, idx_ptr or None, latent_ptr or None, stream or None))

    def train_eval_device(self, leaves_ptr: int, n: int, stats_ptr: int, recon_sums_ptr: int, recon_ptr: int = 0, stream: int = 0):
        self._check(self._lib.vqhip_train_eval_device(self._h, leaves_ptr, n, stats_ptr, recon_sums_ptr, recon_ptr or None, stream or None))

    def train_vq_update_device(self, stats_ptr: int, decay: float = 0.95, eps: float = 1e-4, stream: int = 0):
        self._check(self._lib.vqhip_train_vq_update_device(self._h, stats_ptr, decay, eps, stream or None))

    def train_get_state(self):
        emb, cs, avg = np.empty((256, 128), np.float32), np.empty(256, np.float32), np.empty((256, 128), np.float32)
        self._check(self._lib.vqhip_train_get_state(self._h, emb.ctypes.data, cs.ctypes.data, avg.ctypes.data))
        return {"embedding": emb, "cluster_size": cs, "embed_avg": avg}

    def train_set_state(self, embedding=None, cluster_size=None, embed_avg=None):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float32) for a in (embedding, cluster_size, embed_avg)]
        for a, size, what in zip(arrs, (256 * 128, 256, 256 * 128), ("embedding", "cluster_size", "embed_avg")):
            if a is not None and a.size != size:
                raise ValueError(f"{what}: expected {size} float32 values, got {a.size}")
        self._check(self._lib.vqhip_train_set_state(self._h, *[None if a is None else a.ctypes.data for a in arrs]))

    def train_commit(self):
        self._check(self._lib.vqhip_train_commit(self._h))

    def set_small_batch_tiles(self, tiles: int):
        """-1: automatic choice between the position-split and the one-wave-per-tile path (default); 0: never split;
        n > 0: split passes of up to n tiles (encode) / 1.25 n (decode)."""
        self._check(self._lib.vqhip_set_small_batch_tiles(self._h, tiles))

    # ---- full training step (stage 2) ----
    def fulltrain_begin(self):
        self._check(self._lib.vqhip_fulltrain_begin(self._h))

    def fulltrain_param_count(self) -> int:
        return int(self._lib.vqhip_fulltrain_param_count(self._h))

    def fulltrain_forward_device(self, leaves_ptr: int, n: int, stream: int = 0):
        self._check(self._lib.vqhip_fulltrain_forward_device(self._h, leaves_ptr, n, stream or None))

    def fulltrain_fwdbwd_device(self, leaves_ptr: int, n: int, n_global: int, grads_ptr: int, aux_ptr: int = 0, stream: int = 0):
        self._check(self._lib.vqhip_fulltrain_fwdbwd_device(self._h, leaves_ptr, n, n_global, grads_ptr, aux_ptr or None, stream or None))

    def fulltrain_fwdbwd_overlap_device(self, leaves_ptr: int, n: int, n_global: int, grads_ptr: int, aux_ptr: int, stream: int, decoder_done):
        """fwdbwd with `decoder_done()` called once the decoder half of the backward pass is enqueued (see vqvdb_hip.h)."""
        def cb(_user):
            try:
                decoder_done()
                return 0
            except Exception as e:  # noqa: BLE001 — an exception must not unwind through the C frames
                print(f"fulltrain decoder_done callback: {e}", file=sys.stderr)
                return 1
        fn = PHASE_FN(cb)
        self._check(self._lib.vqhip_fulltrain_fwdbwd_overlap_device(self._h, leaves_ptr, n, n_global, grads_ptr, aux_ptr or None, stream or None, fn, None))

    def fulltrain_set_folded_tail(self, on: bool):
        self._check(self._lib.vqhip_fulltrain_set_folded_tail(self._h, int(on)))

    def fulltrain_decoder_offset(self) -> int:
        return int(self._lib.vqhip_fulltrain_decoder_offset(self._h))

    def fulltrain_ready_stream(self) -> int:
        """Raw HIP stream on which the decoder's gradients are complete at the decoder_done callback (0: the stream of the call)."""
        return int(self._lib.vqhip_fulltrain_ready_stream(self._h) or 0)

    def fulltrain_apply_device(self, grads_ptr: int, aux_ptr: int, lr: float, step: int, betas=(0.9, 0.999), adam_eps: float = 1e-8,
                               weight_decay: float = 1e-4, ema_decay: float = 0.95, ema_eps: float = 1e-4, stream: int = 0):
        self._check(self._lib.vqhip_fulltrain_apply_device(self._h, grads_ptr, aux_ptr or None, lr, step, betas[0], betas[1], adam_eps, weight_decay,
                                                           ema_decay, ema_eps, stream or None))

    def fulltrain_get_params(self) -> np.ndarray:
        out = np.empty(self.fulltrain_param_count(), dtype=np.float32)
        self._check(self._lib.vqhip_fulltrain_get_params(self._h, out.ctypes.data))
        return out

    def fulltrain_set_params(self, flat: np.ndarray):
        flat = np.ascontiguousarray(flat, dtype=np.float32)
        if flat.size != self.fulltrain_param_count():
            raise ValueError("flat parameter vector has the wrong length")
        self._check(self._lib.vqhip_fulltrain_set_params(self._h, flat.ctypes.data))

    def fulltrain_get_opt_state(self):
        """AdamW moments (exp_avg, exp_avg_sq), flat parameter order."""
        m, v = np.empty(self.fulltrain_param_count(), np.float32), np.empty(self.fulltrain_param_count(), np.float32)
        self._check(self._lib.vqhip_fulltrain_get_opt_state(self._h, m.ctypes.data, v.ctypes.data))
        return m, v

    def fulltrain_set_opt_state(self, exp_avg: np.ndarray, exp_avg_sq: np.ndarray):
        m, v = (np.ascontiguousarray(a, dtype=np.float32) for a in (exp_avg, exp_avg_sq))
        if m.size != self.fulltrain_param_count() or v.size != m.size:
            raise ValueError("optimizer moments have the wrong length")
        self._check(self._lib.vqhip_fulltrain_set_opt_state(self._h, m.ctypes.data, v.ctypes.data))

    def fetch(self, name: str, n: int, channels: int, positions: int) -> np.ndarray:
        """debug_fetch without the debug flag: any named workspace tensor as [n, channels, positions]."""
        out = np.empty((n, channels, positions), dtype=np.float32)
        self._check(self._lib.vqhip_debug_fetch(self._h, name.encode(), n, out.ctypes.data))
        return out

    def workspace_bytes(self) -> int:
        return int(self._lib.vqhip_workspace_bytes(self._h))

    def chunk_leaves(self) -> int:
        return int(self._lib.vqhip_chunk_leaves(self._h))

    def reserve(self, n: int):
        self._check(self._lib.vqhip_reserve(self._h, n))

    def profile_enable(self, on: bool):
        self._check(self._lib.vqhip_profile_enable(self._h, int(on)))

    def profile_read(self) -> list:
        stats = (_KernelStat * 256)()
        cnt = ctypes.c_int()
        self._check(self._lib.vqhip_profile_read(self._h, stats, 256, ctypes.byref(cnt)))
        return [dict(name=s.name.decode(), launches=s.launches, total_ms=s.total_ms, flops_per_leaf=s.flops_per_leaf,
                     eff_flops_per_leaf=s.eff_flops_per_leaf, leaves=s.leaves) for s in stats[:cnt.value]]

    def debug_enable(self, on: bool):
        self._check(self._lib.vqhip_debug_enable(self._h, int(on)))

    def debug_fetch(self, name: str, n: int, channels: int, positions: int) -> np.ndarray:
        out = np.empty((n, channels, positions), dtype=np.float32)
        self._check(self._lib.vqhip_debug_fetch(self._h, name.encode(), n, out.ctypes.data))
        return out

    def selftest_mfma(self) -> list:
        out = (ctypes.c_int64 * 2)()
        self._check(self._lib.vqhip_selftest_mfma(self._h, out))
        return list(out)


class HipMultiCodec:
    """In-process multi-GPU front end: contiguous leaf ranges over several devices, no collective."""

    def __init__(self, pack: Union[str, os.PathLike, bytes], device_ids: Sequence[int]):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        ids = (ctypes.c_int * len(device_ids))(*device_ids)
        self._n = len(device_ids)
        if isinstance(pack, (bytes, bytearray, memoryview)):
            self._pack = bytes(pack)
            rc = self._lib.vqhip_multi_create(None, self._pack, len(self._pack), ids, len(device_ids), ctypes.byref(self._h))
        else:
            rc = self._lib.vqhip_multi_create(os.fspath(pack).encode(), None, 0, ids, len(device_ids), ctypes.byref(self._h))
        if rc != 0:
            raise RuntimeError(self._lib.vqhip_multi_last_error(None).decode())

    def _check(self, rc: int):
        if rc != 0:
            raise RuntimeError(self._lib.vqhip_multi_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.vqhip_multi_destroy(self._h)
            self._h = ctypes.c_void_p()

    __del__ = close

    def worker_info(self) -> list:
        """Per device worker: (device_id, numa_node or -1, cores bound or 0)."""
        out = []
        for k in range(self._n):
            d, nn, cb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            self._check(self._lib.vqhip_multi_worker_info(self._h, k, ctypes.byref(d), ctypes.byref(nn), ctypes.byref(cb)))
            out.append((d.value, nn.value, cb.value))
        return out

    def encode(self, leaves: np.ndarray) -> np.ndarray:
        leaves = np.ascontiguousarray(leaves, dtype=np.float32).reshape(-1, LEAF_VOXELS)
        idx = np.empty((leaves.shape[0], LATENT_VOXELS), dtype=np.uint8)
        self._check(self._lib.vqhip_multi_encode(self._h, leaves.ctypes.data, leaves.shape[0], idx.ctypes.data))
        return idx

    def decode(self, indices: np.ndarray) -> np.ndarray:
        indices = np.ascontiguousarray(indices, dtype=np.uint8).reshape(-1, LATENT_VOXELS)
        out = np.empty((indices.shape[0], LEAF_VOXELS), dtype=np.float32)
        self._check(self._lib.vqhip_multi_decode(self._h, indices.ctypes.data, indices.shape[0], out.ctypes.data))
        return out


class IVQVAECodec:
    """Abstract codec (IVQVAECodec.hpp:99-136)."""

    @staticmethod
    def create(config: CodecConfig, type: BackendType) -> Optional["IVQVAECodec"]:
        """Factory (IVQVAECodec.cpp:76-110): any failure is reported on stderr and yields None."""
        try:
            if type == BackendType.HIP:
                return HipBackend(config)
            raise RuntimeError("Requested backend type is not available or disabled in the build configuration.")
        except Exception as e:  # noqa: BLE001 — mirrors catch (const std::exception&)
            print(f"Failed to create VQ-VAE backend: {e}", file=sys.stderr)
            return None

    def encode(self, leafBatch: TensorView) -> Tensor:
        raise NotImplementedError

    def decode(self, indices: TensorView) -> Tensor:
        raise NotImplementedError

    def getLatentShape(self) -> list:
        raise NotImplementedError


class HipBackend(IVQVAECodec):
    """MI355X backend behind the reference's plugin surface (counterpart of TorchBackend,
    src/backends/torch/TorchBackend.cpp)."""

    def __init__(self, config: CodecConfig):
        if config.device != CodecConfig.Device.CUDA:
            raise RuntimeError("HIP backend requires Device::CUDA (GPU); there is no CPU path in this backend")
        source = config.source
        if isinstance(source, EmbeddedModel):
            # the Python side's bin_model.h: a pack installed beside the package (the C++ adapter compiles one in)
            source = os.environ.get("VQVDB_HIP_EMBEDDED_PACK_FILE", EMBEDDED_PACK_FILE)
            if not os.path.exists(source):
                raise RuntimeError("vqhip_create: no weight pack given (embedded model absent from this build)")
        self._codec = HipCodec(source, config.device_id)
        self._latent = self._codec.latent_shape()

    def encode(self, leafBatch: TensorView) -> Tensor:
        if leafBatch.dtype != DataType.FLOAT32:
            raise RuntimeError("encode expects FLOAT32 data.")          # TorchBackend.cpp:134-136
        shape = list(leafBatch.shape)
        if len(shape) != 5 or shape[1:] != [1, 8, 8, 8] or shape[0] < 1:
            raise RuntimeError("encode expects shape [B,1,8,8,8].")
        idx = self._codec.encode(np.asarray(leafBatch.data).reshape(shape[0], LEAF_VOXELS))
        return Tensor(idx.reshape(-1).view(np.uint8), [shape[0]] + self._latent, DataType.UINT8)

    def decode(self, indices: TensorView) -> Tensor:
        if indices.dtype != DataType.UINT8:
            raise RuntimeError("decode expects UINT8 data.")            # TorchBackend.cpp:167-169
        shape = list(indices.shape)
        if len(shape) != 4 or shape[1:] != self._latent or shape[0] < 1:
            raise RuntimeError("decode expects shape [B,4,4,4].")
        out = self._codec.decode(np.asarray(indices.data).reshape(shape[0], LATENT_VOXELS))
        return Tensor(out.reshape(-1).view(np.uint8), [shape[0], 1, 8, 8, 8], DataType.FLOAT32)

    def getLatentShape(self) -> list:
        return list(self._latent)

    @property
    def raw(self) -> HipCodec:
        return self._codec
