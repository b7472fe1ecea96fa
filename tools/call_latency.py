#!/usr/bin/env python3
"""Wall time per host-pointer call (vqhip_encode / vqhip_decode, pageable numpy buffers) at SOP-sized batches, next to the
device-only time of the same batch (encode_device on resident buffers + stream sync).  Run on the GPU box, no torch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec

c = HipCodec(weightpack.dumps(synth.make_weights(0)))
for n in [int(a) for a in sys.argv[1:]] or [64, 256, 1024, 8192]:
    x = synth.make_leaves(n, seed=5)
    idx = np.zeros((n, 64), np.uint8)
    out = np.zeros((n, 512), np.float32)
    for _ in range(20):
        c.encode(x, out=idx)
        c.decode(idx, out=out)
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        c.encode(x, out=idx)
    te = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        c.decode(idx, out=out)
    td = (time.perf_counter() - t0) / reps
    print(f"{n:6d} leaves: encode {te * 1e3:.3f} ms/call, decode {td * 1e3:.3f} ms/call")
