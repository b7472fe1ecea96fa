"""Latency / throughput of encode (and decode) at small batch sizes, position-split path on vs off.  Run on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec

pack = weightpack.dumps(synth.make_weights(0))
codec = HipCodec(pack)
sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536]
x = torch.rand(max(sizes), 512, device="cuda")
idx = torch.empty(max(sizes), 64, dtype=torch.uint8, device="cuda")
rec = torch.empty(max(sizes), 512, device="cuda")


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


print(f"{'leaves':>8s} | {'enc classic':>22s} | {'enc split':>22s} | {'dec classic':>22s} | {'dec split':>22s}")
for n in sizes:
    row = []
    for leg in ("enc", "dec"):
        for tiles in (0, 1 << 20):
            codec.set_small_batch_tiles(tiles)
            if leg == "enc":
                dt = t(lambda: codec.encode_device(x.data_ptr(), n, idx.data_ptr()))
            else:
                dt = t(lambda: codec.decode_device(idx.data_ptr(), n, rec.data_ptr()))
            row.append(f"{dt * 1e3:7.3f} ms {n / dt / 1e6:6.3f} M/s")
    print(f"{n:8d} | " + " | ".join(row))
