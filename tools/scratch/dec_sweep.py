import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec
c = HipCodec(weightpack.dumps(synth.make_weights(0)))
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
for n in [int(a) for a in sys.argv[1:]] or (64, 256, 1024, 2048):
    x = torch.from_numpy(synth.make_leaves(n, seed=3)).to(dev)
    idx = torch.empty((n, 64), dtype=torch.uint8, device=dev); rec = torch.empty((n, 512), dtype=torch.float32, device=dev)
    c.encode_device(x.data_ptr(), n, idx.data_ptr(), s)
    for _ in range(5): c.decode_device(idx.data_ptr(), n, rec.data_ptr(), s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): c.decode_device(idx.data_ptr(), n, rec.data_ptr(), s)
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 100
    c.profile_enable(True)
    for _ in range(20): c.decode_device(idx.data_ptr(), n, rec.data_ptr(), s)
    torch.cuda.synchronize(); st = c.profile_read(); c.profile_enable(False)
    k = {q["name"]: q["total_ms"] / q["launches"] * 1e3 for q in st}
    print(f"n={n}: decode {td*1e3:.3f} ms; conv1 {k.get('dec_res64_conv1_s', 0):.1f} us conv2 {k.get('dec_res64_conv2_s', 0):.1f} us")
