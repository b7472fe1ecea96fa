#!/usr/bin/env python3
"""Differential fuzz of the two launch strategies: random batch sizes through the default policy (position-split kernels,
channel-split variants, tail_small_k ...) against the one-wave-per-tile path on a second handle; indices and voxels must be
bit-identical; the decoder front's two kernels (stem_taps_k / stem_fused_k) against each other on random indices.  Usage: python tools/fuzz_paths.py [seconds]  (run on the GPU box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
pack = weightpack.dumps(synth.make_weights(0))
a, b, f = HipCodec(pack), HipCodec(pack), HipCodec(pack)
os.environ["VQHIP_STEM"] = "gather"   # (read at create) decoder front gathered through the L1 (stem_fused_k) instead of streamed through the LDS ring
g = HipCodec(pack)
del os.environ["VQHIP_STEM"]
g.set_small_batch_tiles(0)
b.set_small_batch_tiles(0)          # one wave per tile everywhere
f.set_small_batch_tiles(1 << 20)    # position-split launches everywhere, also for full 2048-tile chunks
rng = np.random.default_rng(2024)
pool = synth.make_leaves(70000, seed=77)
# adversarial leaves mixed in: zeros, ones, spikes, huge, tiny
pool[:8] = 0.0
pool[8:16] = 1.0
pool[16:24] = 0.0
pool[16:24, 100] = 1e4
pool[24:32] *= 1e-30
t0, it, tot = time.time(), 0, 0
while time.time() - t0 < budget:
    r = rng.random()
    n = int(rng.integers(1, 200)) if r < 0.3 else int(rng.integers(1, 4000)) if r < 0.6 else int(rng.integers(1, 70001))
    start = int(rng.integers(0, 70000 - n + 1))
    x = pool[start:start + n]
    ia, ib = a.encode(x), b.encode(x)
    assert np.array_equal(ia, ib), (it, n, start, "indices")
    ra, rb = a.decode(ia), b.decode(ib)
    assert np.array_equal(ra.view(np.uint32), rb.view(np.uint32)), (it, n, start, "voxels")
    if it % 4 == 0:
        assert np.array_equal(f.encode(x), ib), (it, n, start, "indices, split everywhere")
        assert np.array_equal(f.decode(ib).view(np.uint32), rb.view(np.uint32)), (it, n, start, "voxels, split everywhere")
    # random indices too (codes the encoder never emits next to each other)
    ri = rng.integers(0, 256, size=(n, 64), dtype=np.uint8)
    rb2 = b.decode(ri)
    assert np.array_equal(a.decode(ri).view(np.uint32), rb2.view(np.uint32)), (it, n, "random indices")
    assert np.array_equal(g.decode(ri).view(np.uint32), rb2.view(np.uint32)), (it, n, "random indices, LDS-ring stem vs L1-gather stem")
    it += 1
    tot += n
print(f"fuzz ok: {it} rounds, {tot} leaves, {time.time() - t0:.0f} s")
