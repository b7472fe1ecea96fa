import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec
c = HipCodec(weightpack.dumps(synth.make_weights(0)))
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
for n in [int(a) for a in sys.argv[1:]] or (64, 1024, 4096, 8192):
    x = torch.from_numpy(synth.make_leaves(n, seed=3)).to(dev)
    idx = torch.empty((n, 64), dtype=torch.uint8, device=dev)
    for _ in range(5): c.encode_device(x.data_ptr(), n, idx.data_ptr(), s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): c.encode_device(x.data_ptr(), n, idx.data_ptr(), s)
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 100
    c.profile_enable(True)
    for _ in range(20): c.encode_device(x.data_ptr(), n, idx.data_ptr(), s)
    torch.cuda.synchronize(); st = c.profile_read(); c.profile_enable(False)
    k = {q["name"]: q["total_ms"] / q["launches"] * 1e3 for q in st}
    print(f"n={n}: encode {td*1e3:.3f} ms; down {k.get('enc_down_s', 0):.1f} r32c1 {k.get('enc_res32_conv1_s', 0):.1f} r32c2 {k.get('enc_res32_conv2_s', 0):.1f} us")
