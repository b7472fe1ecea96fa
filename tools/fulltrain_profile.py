"""Per-kernel timing of one full training step (forward + backward + AdamW/EMA) at a given per-rank batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec
from vqvdb_amd.full_training import FullTrainer

for n in [int(a) for a in sys.argv[1:]] or [2048]:
    codec = HipCodec(weightpack.dumps(synth.make_weights(0)))
    tr = FullTrainer(codec)
    x = torch.rand(n, 512, device="cuda")
    for _ in range(2):
        tr.step(x, want_metrics=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        tr.step(x, want_metrics=False)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5
    codec.profile_enable(True)
    for _ in range(3):
        tr.step(x, want_metrics=False)
    torch.cuda.synchronize()
    tot = 0.0
    print(f"--- per-rank batch {n}: {wall * 1e3:.3f} ms/step wall = {n / wall / 1e3:.1f} k leaves/s")
    for st in codec.profile_read():
        ms = st["total_ms"] / st["launches"]
        tot += ms
        if ms > 0.004:
            print(f"  {st['name']:26s} {ms:8.4f} ms")
    print(f"  {'sum of kernels':26s} {tot:8.4f} ms")
    codec.close()
