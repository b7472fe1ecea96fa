#!/usr/bin/env python3
"""GPU bring-up diagnostics: per-layer HIP-vs-oracle comparison, MFMA self-test, quick timings.
Not a pytest file (the gated parity tests are tests/test_gpu_parity.py); run via gpurun:
    python tests/gpu_bringup.py [n_leaves]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import DEBUG_SHAPES, DEC_DEBUG, ENC_DEBUG, Oracle  # noqa: E402
from vqvdb_amd import synth, weightpack  # noqa: E402
from vqvdb_amd.codec import HipCodec  # noqa: E402


def cmp(name, got, want):
    diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
    exact = np.array_equal(got.view(np.uint32) & 0x7FFFFFFF | ((got == 0) * 0), want.view(np.uint32) & 0x7FFFFFFF | ((want == 0) * 0)) \
        or np.array_equal(np.where(got == 0, 0.0, got).view(np.uint32), np.where(want == 0, 0.0, want).view(np.uint32))
    nbad = int((np.where(got == 0, 0.0, got).view(np.uint32) != np.where(want == 0, 0.0, want).view(np.uint32)).sum())
    print(f"  {name:10s} max|d|={diff.max():.3e} rel={diff.max() / max(np.abs(want).max(), 1e-30):.3e} bitexact={exact} nbad={nbad}/{got.size}", flush=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 70
    W = synth.make_weights(0)
    names = [t[0] for t in synth.TENSORS]
    orc = Oracle(W, names)
    codec = HipCodec(weightpack.dumps(W))
    print("latent", codec.latent_shape(), "mfma selftest mismatches", codec.selftest_mfma(), flush=True)
    leaves = np.concatenate([synth.make_leaves(n - 8, 1234), synth.edge_leaves()])
    codec.debug_enable(True)
    t = time.time()
    idx = codec.encode(leaves)
    print(f"encode {n} leaves: {time.time() - t:.3f}s", flush=True)
    oidx, odbg = orc.encode(leaves, threads=8, debug=ENC_DEBUG)
    for name in ENC_DEBUG:
        if name in ("e_x12", "e_z"):
            continue
        c, p = DEBUG_SHAPES[name]
        cmp(name, codec.debug_fetch(name, n, c, p), odbg[name])
    bad = np.nonzero(idx != oidx)
    print(f"  indices: {len(bad[0])} mismatches of {idx.size}", flush=True)
    if len(bad[0]):
        print("   first:", [(int(a), int(b), int(idx[a, b]), int(oidx[a, b])) for a, b in zip(*bad)][:10])
    t = time.time()
    rec = codec.decode(oidx)
    print(f"decode {n} leaves: {time.time() - t:.3f}s", flush=True)
    orec, ddbg = orc.decode(oidx, threads=8, debug=DEC_DEBUG)
    for name in ["d_ystem", "d_d2", "d_y4", "d_x6"]:
        c, p = DEBUG_SHAPES[name]
        cmp(name, codec.debug_fetch(name, n, c, p), ddbg[name])
    cmp("recon", rec, orec)
    print("  recon elementwise rel:", float((np.abs(rec - orec) / np.abs(orec)).max()), flush=True)

    # timing at a larger batch (device-resident path needs torch; here host path incl. PCIe)
    if len(sys.argv) > 2:
        nb = int(sys.argv[2])
        big = synth.make_leaves(nb, 4321)
        codec.debug_enable(False)
        codec.encode(big[:1024])
        codec.profile_enable(True)
        t = time.time(); bi = codec.encode(big); dt = time.time() - t
        print(f"encode {nb}: {dt:.3f}s  {nb / dt:.0f} leaves/s (host path)")
        for s in codec.profile_read():
            tf = s['flops_per_leaf'] * s['leaves'] / (s['total_ms'] * 1e-3) / 1e12 if s['total_ms'] > 0 else 0
            print(f"   {s['name']:20s} {s['launches']:3d} launches {s['total_ms']:9.3f} ms  {tf:7.2f} TF nominal")
        t = time.time(); br = codec.decode(bi); dt = time.time() - t
        print(f"decode {nb}: {dt:.3f}s  {nb / dt:.0f} leaves/s (host path)")
        for s in codec.profile_read():
            tf = s['flops_per_leaf'] * s['leaves'] / (s['total_ms'] * 1e-3) / 1e12 if s['total_ms'] > 0 else 0
            print(f"   {s['name']:20s} {s['launches']:3d} launches {s['total_ms']:9.3f} ms  {tf:7.2f} TF nominal")
        # spot parity on the big batch: first 256 leaves vs oracle
        oi = orc.encode(big[:256], threads=8)
        print("  big-batch idx mismatches (first 256 leaves):", int((oi != bi[:256]).sum()))


if __name__ == "__main__":
    main()
