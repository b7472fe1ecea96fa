"""ctypes loader for the CPU oracle (oracle/vqvae_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg —
never from the product package ``vqvdb_amd``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvqvae_oracle.so")

ENC_DEBUG = ["e_y1", "e_a1", "e_y4", "e_a6", "e_x7", "e_y9", "e_x11", "e_x12", "e_z"]
DEC_DEBUG = ["d_ystem", "d_d2", "d_y4", "d_x6", "d_x7", "d_up", "d_ps", "d_pre"]
DEBUG_SLOTS = ENC_DEBUG + DEC_DEBUG
VQ_STATS_FLOATS = 256 + 256 * 128 + 256 + 1   # counts | dw | per-code squared error | rows
DEBUG_SHAPES = {
    "e_y1": (16, 512), "e_a1": (16, 512), "e_y4": (16, 512), "e_a6": (16, 512), "e_x7": (32, 64),
    "e_y9": (32, 64), "e_x11": (32, 64), "e_x12": (32, 64), "e_z": (128, 64),
    "d_ystem": (64, 64), "d_d2": (64, 64), "d_y4": (64, 64), "d_x6": (64, 64), "d_x7": (64, 64),
    "d_up": (256, 64), "d_ps": (32, 512), "d_pre": (1, 512),
}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vqvae_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvqvae_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Oracle:
    """CPU restatement bound to one weight set (dict name -> fp32 array, synth.TENSORS order)."""

    def __init__(self, weights: dict, tensor_names: list[str]):
        self.lib = ctypes.CDLL(build())
        assert self.lib.vqo_tensor_count() == len(tensor_names)
        assert self.lib.vqo_debug_count() == len(DEBUG_SLOTS)
        self._keep = [np.ascontiguousarray(weights[n], dtype=np.float32) for n in tensor_names]
        self._wptr = (ctypes.c_void_p * len(self._keep))(*[a.ctypes.data for a in self._keep])
        for f in (self.lib.vqo_encode, self.lib.vqo_decode):
            f.restype = ctypes.c_int
            f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                          ctypes.c_void_p, ctypes.c_int]
        self.lib.vqo_encode_ex.restype = ctypes.c_int
        self.lib.vqo_encode_ex.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        self.lib.vqo_decode_ex.restype = ctypes.c_int
        self.lib.vqo_decode_ex.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        self.lib.vqo_expf.restype = ctypes.c_float
        self.lib.vqo_expf.argtypes = [ctypes.c_float]
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        self.lib.vqo_vq_assign.argtypes = [vp, i64, vp, vp, ctypes.c_int]
        self.lib.vqo_vq_stats.argtypes = [vp, vp, i64, vp, vp]
        self.lib.vqo_vq_update.argtypes = [vp, ctypes.c_float, ctypes.c_float, vp, vp, vp]

    def _dbg(self, B, want):
        if not want:
            return None, {}
        bufs, ptrs = {}, (ctypes.c_void_p * len(DEBUG_SLOTS))()
        for i, name in enumerate(DEBUG_SLOTS):
            if name in want:
                c, p = DEBUG_SHAPES[name]
                bufs[name] = np.zeros((B, c, p), dtype=np.float32)
                ptrs[i] = bufs[name].ctypes.data
        return ptrs, bufs

    def encode(self, leaves: np.ndarray, threads: int = 1, debug=(), faithful: bool = False):
        """faithful=True quantizes with the reference's expanded fp32 distance on the materialised latent
        (VQVAE_v2.py:364-366) instead of the folded search the GPU runs; same argmin up to near-ties."""
        leaves = np.ascontiguousarray(leaves, dtype=np.float32).reshape(-1, 512)
        B = leaves.shape[0]
        idx = np.zeros((B, 64), dtype=np.uint8)
        ptrs, bufs = self._dbg(B, set(debug))
        rc = self.lib.vqo_encode_ex(self._wptr, leaves.ctypes.data, B, idx.ctypes.data, ptrs, threads, int(faithful))
        if rc:
            raise RuntimeError("oracle encode failed")
        return (idx, bufs) if debug else idx

    def decode(self, idx: np.ndarray, threads: int = 1, debug=(), unfolded: bool = False, tail_skip_rows: bool = False):
        """unfolded=True runs the decoder tail layer by layer (up_conv -> pixel shuffle -> final) instead
        of the folded composite the GPU uses; both are restatements of VQVAE_v2.py:265-275.
        tail_skip_rows=True: the folded tail without the W-rows outside a voxel's reach in H (structurally zero weights), as the
        GPU's full-chunk kernel runs it; the same bits for finite activations."""
        assert not (unfolded and tail_skip_rows)
        idx = np.ascontiguousarray(idx, dtype=np.uint8).reshape(-1, 64)
        B = idx.shape[0]
        out = np.zeros((B, 512), dtype=np.float32)
        ptrs, bufs = self._dbg(B, set(debug))
        rc = self.lib.vqo_decode_ex(self._wptr, idx.ctypes.data, B, out.ctypes.data, ptrs, threads, 2 if tail_skip_rows else int(unfolded))
        if rc:
            raise RuntimeError("oracle decode failed")
        return (out, bufs) if debug else out

    def expf(self, x: float) -> float:
        return float(self.lib.vqo_expf(ctypes.c_float(x)))

    # ---- codebook training: VectorQuantizerEMA.forward in training mode (VQVAE_v2.py:107-156) ----
    def latent(self, leaves: np.ndarray, threads: int = 1) -> np.ndarray:
        """Encoder output as the reference's `flat` rows: [n*64, 128], row = leaf*64 + position (:113-114)."""
        _, dbg = self.encode(leaves, threads=threads, debug=["e_z"])
        return np.ascontiguousarray(dbg["e_z"].transpose(0, 2, 1)).reshape(-1, 128)

    def vq_assign(self, z: np.ndarray, embedding: np.ndarray, threads: int = 1) -> np.ndarray:
        z = np.ascontiguousarray(z, dtype=np.float32).reshape(-1, 128)
        E = np.ascontiguousarray(embedding, dtype=np.float32).reshape(256, 128)
        idx = np.zeros(z.shape[0], dtype=np.uint8)
        assert self.lib.vqo_vq_assign(z.ctypes.data, z.shape[0], E.ctypes.data, idx.ctypes.data, threads) == 0
        return idx

    def vq_stats(self, z: np.ndarray, idx: np.ndarray, embedding: np.ndarray) -> np.ndarray:
        z = np.ascontiguousarray(z, dtype=np.float32).reshape(-1, 128)
        idx = np.ascontiguousarray(idx, dtype=np.uint8).reshape(-1)
        E = np.ascontiguousarray(embedding, dtype=np.float32).reshape(256, 128)
        stats = np.zeros(VQ_STATS_FLOATS, dtype=np.float32)
        assert self.lib.vqo_vq_stats(z.ctypes.data, idx.ctypes.data, z.shape[0], E.ctypes.data, stats.ctypes.data) == 0
        return stats

    def vq_update(self, stats: np.ndarray, state: dict, decay: float = 0.95, eps: float = 1e-4) -> dict:
        """state = {embedding, cluster_size, embed_avg}; returns the updated copy."""
        new = {k: np.ascontiguousarray(v, dtype=np.float32).copy() for k, v in state.items()}
        stats = np.ascontiguousarray(stats, dtype=np.float32)
        assert self.lib.vqo_vq_update(stats.ctypes.data, decay, eps, new["cluster_size"].ctypes.data, new["embed_avg"].ctypes.data,
                                      new["embedding"].ctypes.data) == 0
        return new
