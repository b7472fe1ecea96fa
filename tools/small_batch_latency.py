"""ms per call on the device at SOP-sized batches (default launch policy), encode and decode; optional per-kernel event times.
Usage: python tools/small_batch_latency.py [--kernels] [sizes ...]   (run on the GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec

args = [a for a in sys.argv[1:] if not a.startswith("--")]
kernels = "--kernels" in sys.argv
graph = "--graph" in sys.argv      # also: the same call captured once in a HIP graph and replayed (does the launch chain cost anything?)
sizes = [int(a) for a in args] or [64, 256, 1024, 2048, 8192]
codec = HipCodec(weightpack.dumps(synth.make_weights(0)))
x = torch.rand(max(sizes), 512, device="cuda")
idx = torch.empty(max(sizes), 64, dtype=torch.uint8, device="cuda")
rec = torch.empty(max(sizes), 512, device="cuda")


def t(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for n in sizes:
    enc = lambda: codec.encode_device(x.data_ptr(), n, idx.data_ptr())
    dec = lambda: codec.decode_device(idx.data_ptr(), n, rec.data_ptr())
    te, td = t(enc), t(dec)
    print(f"n={n}: encode {te * 1e3:.4f} ms ({n / te / 1e6:.3f} M/s)  decode {td * 1e3:.4f} ms ({n / td / 1e6:.3f} M/s)")
    if graph:
        side = torch.cuda.Stream()
        res = []
        for fn_of in (lambda st: codec.encode_device(x.data_ptr(), n, idx.data_ptr(), st), lambda st: codec.decode_device(idx.data_ptr(), n, rec.data_ptr(), st)):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                fn_of(side.cuda_stream)
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=side):
                    fn_of(side.cuda_stream)
            torch.cuda.synchronize()
            res.append(t(g.replay))
        print(f"      replayed from a HIP graph: encode {res[0] * 1e3:.4f} ms  decode {res[1] * 1e3:.4f} ms")
    if kernels:
        for name, fn in (("encode", enc), ("decode", dec)):
            codec.profile_enable(True)
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            st = codec.profile_read()
            codec.profile_enable(False)
            print(f"  {name}: {len(st)} kernels, sum of event times {sum(s['total_ms'] / s['launches'] for s in st) * 1e3:.1f} us")
            for s in st:
                print(f"    {s['name']:30s} {s['total_ms'] / s['launches'] * 1e3:7.1f} us")
