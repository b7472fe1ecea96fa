import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec
pack = weightpack.dumps(synth.make_weights(0))
dev = torch.device("cuda:0")
N = 65536
x = torch.from_numpy(synth.make_leaves(N, seed=3)).to(dev)
idx = torch.empty((N, 64), dtype=torch.uint8, device=dev)
rec = torch.empty((N, 512), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
cs = {"split": HipCodec(pack), "large": HipCodec(pack)}
cs["split"].set_small_batch_tiles(1 << 20)
cs["large"].set_small_batch_tiles(0)
for c in cs.values():
    c.encode_device(x.data_ptr(), N, idx.data_ptr(), s); c.decode_device(idx.data_ptr(), N, rec.data_ptr(), s)
torch.cuda.synchronize()
def t(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for n in (40960, 45056, 49152, 53248, 57344, 61440, 65536):
    reps = max(4, 200000 // n)
    row = [f"n={n:6d}"]
    for name, c in cs.items():
        te = t(lambda: c.encode_device(x.data_ptr(), n, idx.data_ptr(), s), reps)
        td = t(lambda: c.decode_device(idx.data_ptr(), n, rec.data_ptr(), s), reps)
        row.append(f"{name}: enc {te*1e3:7.3f} ms ({n/te/1e6:5.2f} M/s) dec {td*1e3:7.3f} ms ({n/td/1e6:5.2f} M/s)")
    print("  ".join(row))
