"""Folded decoder tail of full chunks: the D x H zero-skipping kernel (tail_rows16_k, default) against the slab kernel
(VQHIP_TAIL=slab) and the oracle, bit for bit, then per-kernel decode timings of both at 65 536 leaves.

    python tools/tail_variant_check.py [--leaves 65536] [--steps 10] [--skip-oracle]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqvdb_amd import synth, weightpack  # noqa: E402
from vqvdb_amd.codec import HipCodec  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leaves", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--skip-oracle", action="store_true")
    a = ap.parse_args()
    import torch
    weights = synth.make_weights(seed=0)
    pack = weightpack.dumps(weights)
    rng = np.random.default_rng(5)
    idx = rng.integers(0, 256, size=(a.leaves, 64), dtype=np.uint8)
    outs = {}
    for variant in ("rows16", "groups", "rows32", "slab"):
        os.environ["VQHIP_TAIL"] = variant
        c = HipCodec(pack)
        c.set_small_batch_tiles(0)
        # ragged sizes too: a last tile with one leaf, a half tile boundary
        small = {n: c.decode(idx[:n]) for n in (1, 16, 17, 33, 300, 4097)}
        rec = c.decode(idx)
        outs[variant] = (small, rec)
        d_idx = torch.from_numpy(idx).cuda()
        d_out = torch.empty((a.leaves, 512), dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            c.decode_device(d_idx.data_ptr(), a.leaves, d_out.data_ptr(), st)
        torch.cuda.synchronize()
        c.profile_enable(True)
        for _ in range(a.steps):
            c.decode_device(d_idx.data_ptr(), a.leaves, d_out.data_ptr(), st)
        torch.cuda.synchronize()
        for s in c.profile_read():
            print(f"{variant:5s} {s['name']:20s} {s['total_ms'] / max(s['launches'], 1):8.4f} ms  x{s['launches']}")
        c.profile_enable(False)
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), rec.view(np.uint32))
        c.close()
    (sb, rb) = outs["slab"]
    bad = 0
    for v in ("groups", "rows16", "rows32"):
        (sa, ra) = outs[v]
        b = int((ra.view(np.uint32) != rb.view(np.uint32)).sum())
        print(v, "vs slab, full batch: differing words", b, "of", ra.size)
        bad += b
        for n in sa:
            b = int((sa[n].view(np.uint32) != sb[n].view(np.uint32)).sum())
            print(f"{v} vs slab, n={n}: differing words {b}")
            bad += b
    ra = outs["rows16"][1]
    if not a.skip_oracle:
        from oracle.oracle import Oracle
        o = Oracle(weights, [t[0] for t in synth.TENSORS])
        ref = o.decode(idx[:256], threads=16)
        b = int((ra[:256].view(np.uint32) != ref.view(np.uint32)).sum())
        print("rows vs oracle, 256 leaves: differing words", b)
        bad += b
    print("OK" if bad == 0 else "MISMATCH")
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
