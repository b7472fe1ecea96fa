import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from oracle.oracle import DEBUG_SHAPES, ENC_DEBUG, Oracle
from vqvdb_amd import synth, weightpack
from vqvdb_amd.codec import HipCodec
w = synth.make_weights(seed=0)
orc = Oracle(w, [t[0] for t in synth.TENSORS])
codec = HipCodec(weightpack.dumps(w))
leaves = np.concatenate([synth.make_leaves(120, seed=1234), synth.edge_leaves()])
codec.debug_enable(True)
idx = codec.encode(leaves)
oidx, dbg = orc.encode(leaves, threads=8, debug=ENC_DEBUG)
for name in sys.argv[1:] or ["e_x11"]:
    c, p = DEBUG_SHAPES[name]
    g = codec.debug_fetch(name, len(leaves), c, p)
    o = dbg[name]
    bad = np.argwhere(g != o)
    print(name, g.shape, "mismatches", len(bad), "of", g.size)
    if len(bad):
        print(" leaves", np.unique(bad[:, 0])[:40], "\n channels", np.unique(bad[:, 1]), "\n positions", np.unique(bad[:, 2]))
        for b in bad[:6]:
            print("  ", b, g[tuple(b)], o[tuple(b)])
