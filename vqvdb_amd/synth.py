"""Deterministic synthetic weights and leaves for the VQ-VAE leaf codec.

The reference snapshot ships no trained weights (``/root/reference/.MISSING_LARGE_BLOBS``),
so parity and throughput are pinned on *synthetic* tensors.  They come from a
counter-based generator (SplitMix64 finaliser over ``(seed, tensor_id, element)``)
so that this container, the GPU box and the C oracle all regenerate the same
bits from nothing but numpy — no fixture file has to carry the weights.

Tensor names and shapes are the ``state_dict()`` of the reference model
``VQVAE(1, 128, 256, 0.25)`` (python/VQVAE_v2.py:328-343; shapes SURVEY.md App. A-2).
Scales are "trained-like" (every conv O(1/sqrt(fan_in)), non-identity GroupNorm
affine) because the reference's default init makes residual branches ~1e-4 of
the signal (VQVAE_v2.py:201-202) and would hide bugs.
"""
from __future__ import annotations

import numpy as np

K_CODES = 256   # quantizer.embedding rows   (training.py:53)
D_EMBED = 128   # embedding dim              (training.py:54)

# (name, shape, kind)   kind: conv | bias | gn_w | gn_b | fc | codebook
TENSORS = [
    ("encoder.pre.0.weight", (16, 1, 3, 3, 3), "conv"),
    ("encoder.pre.0.bias", (16,), "bias"),
    ("encoder.pre.1.weight", (16,), "gn_w"),
    ("encoder.pre.1.bias", (16,), "gn_b"),
    ("encoder.pre.3.gn1.weight", (16,), "gn_w"),
    ("encoder.pre.3.gn1.bias", (16,), "gn_b"),
    ("encoder.pre.3.conv1.weight", (16, 16, 3, 3, 3), "conv"),
    ("encoder.pre.3.conv1.bias", (16,), "bias"),
    ("encoder.pre.3.gn2.weight", (16,), "gn_w"),
    ("encoder.pre.3.gn2.bias", (16,), "gn_b"),
    ("encoder.pre.3.conv2.weight", (16, 16, 3, 3, 3), "conv"),
    ("encoder.pre.3.conv2.bias", (16,), "bias"),
    ("encoder.down.weight", (32, 16, 4, 4, 4), "conv"),
    ("encoder.down.bias", (32,), "bias"),
    ("encoder.res_stack.0.gn1.weight", (32,), "gn_w"),
    ("encoder.res_stack.0.gn1.bias", (32,), "gn_b"),
    ("encoder.res_stack.0.conv1.weight", (32, 32, 3, 3, 3), "conv"),
    ("encoder.res_stack.0.conv1.bias", (32,), "bias"),
    ("encoder.res_stack.0.gn2.weight", (32,), "gn_w"),
    ("encoder.res_stack.0.gn2.bias", (32,), "gn_b"),
    ("encoder.res_stack.0.conv2.weight", (32, 32, 3, 3, 3), "conv"),
    ("encoder.res_stack.0.conv2.bias", (32,), "bias"),
    ("encoder.attn.fc.0.weight", (8, 32), "fc"),
    ("encoder.attn.fc.2.weight", (32, 8), "fc"),
    ("encoder.proj.weight", (128, 32, 1, 1, 1), "conv"),
    ("encoder.proj.bias", (128,), "bias"),
    ("decoder.stem.0.weight", (64, 128, 3, 3, 3), "conv"),
    ("decoder.stem.0.bias", (64,), "bias"),
    ("decoder.stem.1.weight", (64,), "gn_w"),
    ("decoder.stem.1.bias", (64,), "gn_b"),
    ("decoder.res_stack.0.gn1.weight", (64,), "gn_w"),
    ("decoder.res_stack.0.gn1.bias", (64,), "gn_b"),
    ("decoder.res_stack.0.conv1.weight", (64, 64, 3, 3, 3), "conv"),
    ("decoder.res_stack.0.conv1.bias", (64,), "bias"),
    ("decoder.res_stack.0.gn2.weight", (64,), "gn_w"),
    ("decoder.res_stack.0.gn2.bias", (64,), "gn_b"),
    ("decoder.res_stack.0.conv2.weight", (64, 64, 3, 3, 3), "conv"),
    ("decoder.res_stack.0.conv2.bias", (64,), "bias"),
    ("decoder.attn.fc.0.weight", (16, 64), "fc"),
    ("decoder.attn.fc.2.weight", (64, 16), "fc"),
    ("decoder.up_conv.weight", (256, 64, 3, 3, 3), "conv"),
    ("decoder.up_conv.bias", (256,), "bias"),
    ("decoder.final.weight", (1, 32, 3, 3, 3), "conv"),
    ("decoder.final.bias", (1,), "bias"),
    ("quantizer.embedding", (K_CODES, D_EMBED), "codebook"),
]

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """SplitMix64 output function on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def uniform01(seed: int, stream: int, n: int, start: int = 0) -> np.ndarray:
    """``n`` floats in [0,1) with 24 random bits each; element ``i`` depends only on
    ``(seed, stream, start+i)`` so any slice can be regenerated independently."""
    idx = np.arange(start, start + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = (np.uint64(seed) * np.uint64(0xD1342543DE82EF95)
               + (np.uint64(stream) << np.uint64(40))) & _M64
        bits = _splitmix64(idx + key)
    return ((bits >> np.uint64(40)).astype(np.float32)
            * np.float32(1.0 / (1 << 24))).astype(np.float32)


def make_weights(seed: int = 0) -> dict[str, np.ndarray]:
    """Synthetic fp32 parameter set with the reference model's names and shapes."""
    out: dict[str, np.ndarray] = {}
    for tid, (name, shape, kind) in enumerate(TENSORS):
        n = int(np.prod(shape))
        u = uniform01(seed, tid + 1, n)
        if kind == "conv":
            fan_in = int(np.prod(shape[1:]))
            a = np.float32(np.sqrt(3.0 / fan_in))
            v = (u * np.float32(2.0) - np.float32(1.0)) * a
        elif kind == "fc":
            a = np.float32(np.sqrt(3.0 / shape[1]))
            v = (u * np.float32(2.0) - np.float32(1.0)) * a
        elif kind == "bias":
            v = (u * np.float32(2.0) - np.float32(1.0)) * np.float32(0.1)
        elif kind == "gn_w":
            v = np.float32(1.0) + (u * np.float32(2.0) - np.float32(1.0)) * np.float32(0.3)
        elif kind == "gn_b":
            v = (u * np.float32(2.0) - np.float32(1.0)) * np.float32(0.2)
        elif kind == "codebook":
            # Irwin-Hall(4) pseudo-normal, scaled to the encoder's latent spread so
            # the nearest-code search is non-trivial (many codes in use).
            u4 = uniform01(seed, tid + 1, 4 * n).reshape(n, 4).sum(axis=1, dtype=np.float32)
            v = (u4 - np.float32(2.0)) * np.float32(0.35)
        else:  # pragma: no cover
            raise ValueError(kind)
        out[name] = np.ascontiguousarray(v.astype(np.float32).reshape(shape))
    return out


def make_leaves(n: int, seed: int = 1234, start: int = 0) -> np.ndarray:
    """``n`` leaves of 512 uniform-[0,1) voxels, leaf-major ``[n, 512]`` — the
    layout ``VDBInputBlockStreamer::nextBatch`` hands the backend
    (src/orchestrator/VQVAECodec.cpp:36-59).  Leaf ``i`` depends only on
    ``(seed, start+i)``."""
    return uniform01(seed, 0, n * 512, start * 512).reshape(n, 512)


def edge_leaves() -> np.ndarray:
    """Edge-case leaves: all-zero, all-one, single spike,
    negative values, large values, a d/h/w ramp (transpose detector), constant 0.5."""
    e = np.zeros((8, 512), dtype=np.float32)
    e[1] = 1.0
    e[2, 3 * 64 + 4 * 8 + 5] = 1.0
    e[3] = -make_leaves(1, seed=77)[0]
    e[4] = make_leaves(1, seed=78)[0] * np.float32(50.0)
    d, h, w = np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij")
    e[5] = ((d * 1.0 + h * 0.1 + w * 0.01) / 8.0).astype(np.float32).reshape(512)
    e[6] = 0.5
    e[7] = make_leaves(1, seed=79)[0] * np.float32(1e-3)
    return e


def sparse_leaves(n: int, seed: int = 2468) -> np.ndarray:
    """``n`` leaves shaped like real VDB content rather than white noise ("real VDB leaves are sparse /
    background-dominated"): mostly-background leaves with a smooth blob (fog-volume falloff), a narrow-band ramp across a
    plane (level-set like, saturating at 0 / 1), background with a handful of active voxels, shifted and x5-scaled noise
    (values outside [0,1]), near-constant leaves and exact zeros.  Leaf ``i`` depends only on ``(seed, i)``; float32
    arithmetic with correctly rounded operations only (+ - * / sqrt, min / max), so it regenerates bit-identically."""
    f = np.float32
    u = uniform01(seed, 7, n * 16).reshape(n, 16)          # per-leaf parameters
    noise = uniform01(seed, 8, n * 512).reshape(n, 512)
    d, h, w = np.meshgrid(np.arange(8, dtype=np.float32), np.arange(8, dtype=np.float32), np.arange(8, dtype=np.float32), indexing="ij")
    P = np.stack([d.reshape(-1), h.reshape(-1), w.reshape(-1)], axis=0)      # [3,512], offset d*64+h*8+w
    out = np.zeros((n, 512), dtype=np.float32)
    kind = u[:, 0]
    for i in range(n):
        k = kind[i]
        if k < f(0.40):      # blob: clamp(1 - |p-c|/r, 0, 1) * amplitude, centre possibly outside the leaf
            c = (u[i, 1:4] * f(14.0) - f(3.0)).reshape(3, 1)
            r = f(2.0) + u[i, 4] * f(8.0)
            dist = np.sqrt(((P - c) * (P - c)).sum(axis=0, dtype=np.float32))
            out[i] = np.maximum(f(1.0) - dist / r, f(0.0)) * (f(0.25) + f(0.75) * u[i, 5])
        elif k < f(0.60):    # narrow band across a plane: clamp(0.5 + (n.p - d)/w, 0, 1)
            nv = (u[i, 1:4] * f(2.0) - f(1.0)).reshape(3, 1)
            off = u[i, 4] * f(10.0) - f(1.5)
            wd = f(0.75) + u[i, 5] * f(3.0)
            out[i] = np.minimum(np.maximum(f(0.5) + ((nv * P).sum(axis=0, dtype=np.float32) - off) / wd, f(0.0)), f(1.0))
        elif k < f(0.75):    # background + a handful of active voxels
            cnt = 1 + int(u[i, 1] * f(8.0))
            pos = (uniform01(seed, 9, cnt, start=i * 8) * f(512.0)).astype(np.int64) % 512
            out[i, pos] = noise[i, :cnt]
        elif k < f(0.85):    # shifted, x5-scaled noise: values in [-1, 4)
            out[i] = noise[i] * f(5.0) - f(1.0)
        elif k < f(0.95):    # near-constant leaf
            out[i] = u[i, 1] + noise[i] * f(1e-4)
        # else: exact zeros (pure background)
    return out
