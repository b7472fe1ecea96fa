"""Times the host-pointer C ABI (vqhip_encode / vqhip_decode) with fresh vs reused caller buffers, and the raw
pinned / pageable copy rates, to see what bounds the PCIe-inclusive numbers.  Run on the GPU box."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec

B = 65536
N = 8 * B
codec = HipCodec(weightpack.dumps(synth.make_weights(0)))
leaves = np.tile(synth.make_leaves(B, seed=1234), (8, 1))
idx = codec.encode(leaves)


def t(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    t.last = sorted(ts)
    return t.last[0]


def both(f):
    best = t(f)
    return f"{N / best / 1e6:.2f} M leaves/s best, {N / t.last[len(t.last) // 2] / 1e6:.2f} median of 5"


out = np.zeros((N, 512), np.float32)
iout = np.zeros((N, 64), np.uint8)
print(f"encode fresh out      : {both(lambda: codec.encode(leaves))}")
print(f"encode reused out     : {both(lambda: codec.encode(leaves, out=iout))}")
print(f"decode fresh out      : {both(lambda: codec.decode(idx))}")
print(f"decode reused out     : {both(lambda: codec.decode(idx, out=out))}")
# raw copies
try:  # after the timed calls: a process that imports torch first binds this library to the wheel's bundled HIP runtime
    import torch
    torch.zeros(1).cuda()
except Exception as e:  # noqa: BLE001 — two HIP runtimes in one process: the second one finds no device
    print("raw copy rates skipped:", e)
    sys.exit(0)
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
