import ctypes, sys, time, os
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.zeros(1).cuda()
sys.path.insert(0, os.getcwd())
from vqvdb_amd.codec import HipCodec  # loads libvqvdb_hip.so -> whichever libamdhip64 resolves
from vqvdb_amd import synth, weightpack
c = HipCodec(weightpack.dumps(synth.make_weights(0)))
maps = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l]
print(sorted(set(maps)))
hip = ctypes.CDLL([m for m in maps if "libamdhip64" in m][0])
n = 128 << 20
p = ctypes.c_void_p()
assert hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(n), 0) == 0
a = np.ones(n, np.uint8); b = np.zeros(n, np.uint8)
for name, dst, src in (("pageable->pinned", p.value, a.ctypes.data), ("pinned->pageable", b.ctypes.data, p.value), ("pageable->pageable", b.ctypes.data, a.ctypes.data)):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); ctypes.memmove(dst, src, n); best = min(best, time.perf_counter() - t0)
    print(f"{name}: {n / best / 1e9:.1f} GB/s (1 thread)")
