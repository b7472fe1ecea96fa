import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec
c = HipCodec(weightpack.dumps(synth.make_weights(0)))
dev = torch.device("cuda:0"); n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.from_numpy(synth.make_leaves(n, seed=3)).to(dev)
idx = torch.empty((n, 64), dtype=torch.uint8, device=dev); rec = torch.empty((n, 512), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(10): c.encode_device(x.data_ptr(), n, idx.data_ptr(), s); c.decode_device(idx.data_ptr(), n, rec.data_ptr(), s)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): c.encode_device(x.data_ptr(), n, idx.data_ptr(), s)
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 200; t0 = time.perf_counter()
for _ in range(200): c.decode_device(idx.data_ptr(), n, rec.data_ptr(), s)
torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 200
print(f"n={n}: encode {te*1e6:.1f} us, decode {td*1e6:.1f} us per call")
