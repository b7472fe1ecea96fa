"""Random batch sizes through the full training step (forward + backward) of three builds of the same codec: default (rolling-window /
plane-walking weight gradients, two streams), VQHIP_TRAIN_STREAMS=1 (one stream: must give the SAME BITS) and VQHIP_TRAIN_WGRAD=pairs
(the pair-per-step weight-gradient kernels: same sums in another order, <= 2e-5 of each tensor's maximum; everything else the same bits).
Usage: python tools/fuzz_training.py [seconds]  (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
weights = synth.make_weights(0)
pack = weightpack.dumps(weights)


def make(env):
    for k in ("VQHIP_TRAIN_STREAMS", "VQHIP_TRAIN_WGRAD"):
        os.environ.pop(k, None)
    os.environ.update(env)
    c = HipCodec(pack)
    c.fulltrain_begin()
    return c


two, one, pairs = make({}), make({"VQHIP_TRAIN_STREAMS": "1"}), make({"VQHIP_TRAIN_WGRAD": "pairs"})
os.environ.pop("VQHIP_TRAIN_WGRAD", None)
conv = []   # slices of the flat vector that hold conv weights (5-D tensors)
off = 0
for name, shape, _ in synth.TENSORS:
    if name.startswith("quantizer."):
        continue
    n = int(np.prod(shape))
    if len(shape) == 5:
        conv.append((name, off, off + n))
    off += n
mask = np.zeros(off, bool)
for _, a, b in conv:
    mask[a:b] = True
rng = np.random.default_rng(2026)
t0, rounds, leaves, worst = time.time(), 0, 0, 0.0
while time.time() - t0 < budget:
    n = int(rng.choice([rng.integers(1, 97), rng.integers(97, 1025), rng.integers(1025, 3073)]))
    x = torch.from_numpy(synth.make_leaves(n, seed=int(rng.integers(1 << 30)))).cuda()
    G = [torch.zeros(off, device="cuda") for _ in range(3)]
    for c, g in zip((two, one, pairs), G):
        c.fulltrain_fwdbwd_device(x.data_ptr(), n, n, g.data_ptr())
    torch.cuda.synchronize()
    g2, g1, gp = (g.cpu().numpy() for g in G)
    assert np.array_equal(g2.view(np.uint32), g1.view(np.uint32)), f"n={n}: two streams differ from one"
    assert np.array_equal(g2[~mask].view(np.uint32), gp[~mask].view(np.uint32)), f"n={n}: non-conv gradients differ between the kernel families"
    for name, a, b in conv:
        e = float(np.abs(g2[a:b] - gp[a:b]).max() / max(float(np.abs(gp[a:b]).max()), 1e-30))
        worst = max(worst, e)
        assert e < 2e-5, f"n={n}: {name} differs by {e:.2e} between the kernel families"
    rounds += 1
    leaves += n
print(f"training fuzz ok: {rounds} rounds, {leaves} leaves, worst conv-weight difference between the families {worst:.1e}, {time.time() - t0:.0f} s")
