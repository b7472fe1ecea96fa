"""Times the host-pointer C ABI (vqhip_encode / vqhip_decode) with fresh vs reused caller buffers, and the raw
pinned / pageable copy rates, to see what bounds the PCIe-inclusive numbers.  Run on the GPU box."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec

B = 65536
N = 8 * B
codec = HipCodec(weightpack.dumps(synth.make_weights(0)))
leaves = np.tile(synth.make_leaves(B, seed=1234), (8, 1))
idx = codec.encode(leaves)


def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best


out = np.zeros((N, 512), np.float32)
iout = np.zeros((N, 64), np.uint8)
print(f"encode fresh out      : {N / t(lambda: codec.encode(leaves)) / 1e6:.2f} M leaves/s")
print(f"encode reused out     : {N / t(lambda: codec.encode(leaves, out=iout)) / 1e6:.2f} M leaves/s")
print(f"decode fresh out      : {N / t(lambda: codec.decode(idx)) / 1e6:.2f} M leaves/s")
print(f"decode reused out     : {N / t(lambda: codec.decode(idx, out=out)) / 1e6:.2f} M leaves/s")
# raw copies
pin = torch.empty(B * 512, dtype=torch.float32).pin_memory()
dev = torch.empty(B * 512, dtype=torch.float32, device="cuda")
def d2h():
    pin.copy_(dev, non_blocking=True); torch.cuda.synchronize()
def h2d():
    dev.copy_(pin, non_blocking=True); torch.cuda.synchronize()
print(f"pinned D2H 128 MiB    : {B * 2048 / t(d2h, 5) / 1e9:.1f} GB/s")
print(f"pinned H2D 128 MiB    : {B * 2048 / t(h2d, 5) / 1e9:.1f} GB/s")
a = np.zeros(B * 512, np.float32); b = np.ones(B * 512, np.float32)
print(f"1-thread memcpy 128MiB: {B * 2048 / t(lambda: np.copyto(a, b), 5) / 1e9:.1f} GB/s")
