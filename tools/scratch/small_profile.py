import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec
c = HipCodec(weightpack.dumps(synth.make_weights(0)))
dev = torch.device("cuda:0")
for n in (64, 1024):
    x = torch.from_numpy(synth.make_leaves(n, seed=3)).to(dev)
    idx = torch.empty((n, 64), dtype=torch.uint8, device=dev)
    rec = torch.empty((n, 512), dtype=torch.float32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        c.encode_device(x.data_ptr(), n, idx.data_ptr(), s); c.decode_device(idx.data_ptr(), n, rec.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): c.encode_device(x.data_ptr(), n, idx.data_ptr(), s)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(50): c.decode_device(idx.data_ptr(), n, rec.data_ptr(), s)
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 50
    c.profile_enable(True)
    for _ in range(20): c.encode_device(x.data_ptr(), n, idx.data_ptr(), s)
    torch.cuda.synchronize()
    st = c.profile_read(); c.profile_enable(False)
    print(f"n={n}: encode {te*1e3:.3f} ms wall, {len(st)} kernels, sum of kernel times {sum(k['total_ms']/k['launches'] for k in st):.3f} ms")
    for k in st: print(f"    {k['name']:28s} {k['total_ms']/k['launches']*1e3:7.1f} us")
    c.profile_enable(True)
    for _ in range(20): c.decode_device(idx.data_ptr(), n, rec.data_ptr(), s)
    torch.cuda.synchronize()
    st = c.profile_read(); c.profile_enable(False)
    print(f"n={n}: decode {td*1e3:.3f} ms wall, {len(st)} kernels, sum of kernel times {sum(k['total_ms']/k['launches'] for k in st):.3f} ms")
    for k in st: print(f"    {k['name']:28s} {k['total_ms']/k['launches']*1e3:7.1f} us")
