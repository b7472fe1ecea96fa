import sys, os, time, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec
pack = weightpack.dumps(synth.make_weights(0))
dev = torch.device("cuda:0")
c = HipCodec(pack)
s = torch.cuda.current_stream().cuda_stream
for n in (64, 256, 1024, 2048, 4096, 8192):
    x = torch.from_numpy(synth.make_leaves(n, seed=3)).to(dev)
    idx = torch.empty((n, 64), dtype=torch.uint8, device=dev); rec = torch.empty((n, 512), dtype=torch.float32, device=dev)
    c.encode_device(x.data_ptr(), n, idx.data_ptr(), s)
    for _ in range(5): c.decode_device(idx.data_ptr(), n, rec.data_ptr(), s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): c.decode_device(idx.data_ptr(), n, rec.data_ptr(), s)
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 50
    c.profile_enable(True)
    for _ in range(20): c.decode_device(idx.data_ptr(), n, rec.data_ptr(), s)
    torch.cuda.synchronize(); st = c.profile_read(); c.profile_enable(False)
    tail = [k for k in st if k["name"] == "dec_tail_s"][0]
    print(f"n={n}: decode {td*1e3:.3f} ms, dec_tail_s {tail['total_ms']/tail['launches']*1e3:.1f} us")
