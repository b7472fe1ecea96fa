"""Soak: many mixed calls on one handle; device memory must stay flat and results must repeat bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec
from vqvdb_amd.full_training import FullTrainer

c = HipCodec(weightpack.dumps(synth.make_weights(0)))
x = synth.make_leaves(3000, seed=1)
ref_idx = c.encode(x); ref_rec = c.decode(ref_idx)
tr = FullTrainer(c)
xb = torch.from_numpy(synth.make_leaves(512, seed=2)).cuda()
free0 = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    n = [5, 3000, 64, 1200][it % 4]
    idx = c.encode(x[:n]); rec = c.decode(idx)
    if it < 4 or it % 50 == 0:
        pass
    tr.evaluate(xb)
    if it == 8:
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("device memory drift (MB):", (free0 - free1) / 1e6)
# determinism of training: two fresh runs give identical parameters
def run():
    cc = HipCodec(weightpack.dumps(synth.make_weights(0)))
    t = FullTrainer(cc)
    for s in range(5):
        t.step(torch.from_numpy(synth.make_leaves(256, seed=10 + s)).cuda(), want_metrics=False)
    p = cc.fulltrain_get_params(); cc.close(); return p
a, b = run(), run()
print("training bitwise reproducible:", bool(np.array_equal(a, b)))
assert abs(free0 - free1) < 64e6 and np.array_equal(a, b)
print("soak ok")
