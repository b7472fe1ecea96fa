"""Per-kernel timing of one codebook-training step (HIP events on the launch stream) at a given per-rank batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec
from vqvdb_amd.codebook_training import CodebookTrainer

for n in [int(a) for a in sys.argv[1:]] or [2048, 65536]:
    codec = HipCodec(weightpack.dumps(synth.make_weights(0)))
    tr = CodebookTrainer(codec)
    x = torch.rand(n, 512, device="cuda")
    for _ in range(2):
        tr.step(x, want_metrics=False)
    torch.cuda.synchronize()
    codec.profile_enable(True)
    for _ in range(4):
        tr.step(x, want_metrics=False)
    torch.cuda.synchronize()
    tot = 0.0
    print(f"--- per-rank batch {n}")
    for st in codec.profile_read():
        ms = st["total_ms"] / st["launches"]
        tot += ms
        print(f"  {st['name']:26s} {ms:8.4f} ms")
    print(f"  {'sum':26s} {tot:8.4f} ms  -> {n / tot / 1e3:.3f} M leaves/s")
    codec.close()
