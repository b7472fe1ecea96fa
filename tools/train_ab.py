#!/usr/bin/env python3
"""Wall time of the full training step (2048 and 8192 leaves per rank) for A/B runs of arrangements selected by environment variables:
    VQHIP_TRAIN_BIAS=main VQHIP_TRAIN_EMA_AT=backward python tools/train_ab.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec
from vqvdb_amd.full_training import FullTrainer

tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("VQHIP_")) or "default"
out = []
for n in ([int(a) for a in sys.argv[1:]] or [2048, 8192]):
    codec = HipCodec(weightpack.dumps(synth.make_weights(0)))
    tr = FullTrainer(codec)
    x = torch.rand(n, 512, device="cuda")
    for _ in range(4):
        tr.step(x, want_metrics=False)
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            tr.step(x, want_metrics=False)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 8)
    m = tr.step(x)
    out.append(f"{n}: {best * 1e3:.3f} ms (loss {m['loss']:.6f})")
    codec.close()
print(f"[{tag}] " + "   ".join(out))
