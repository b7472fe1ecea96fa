#!/usr/bin/env python3
"""GPU-side timeline of the host-pointer entry points (vqhip_encode / vqhip_decode over 8 chunks of 65536 leaves):

    cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/hp -o hp -- python tools/host_pipeline_trace.py run
    python tools/host_pipeline_trace.py show $OUT/hp/hp_results.db

'show' prints, for the last encode and the last decode call, each chunk's H2D / kernels / D2H intervals relative to the first
event of the call, so pipeline bubbles (compute waiting for a copy, copies waiting for the host) are visible.
"""
import os
import sqlite3
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import numpy as np
    from vqvdb_amd import synth, weightpack
    from vqvdb_amd.codec import HipCodec
    B = 65536
    codec = HipCodec(weightpack.dumps(synth.make_weights(0)))
    leaves = np.tile(synth.make_leaves(B, seed=1234), (8, 1))
    iout = np.zeros((8 * B, 64), np.uint8)
    out = np.zeros((8 * B, 512), np.float32)
    for _ in range(2):
        codec.encode(leaves, out=iout)
        codec.decode(iout, out=out)
    for f in (lambda: codec.encode(leaves, out=iout), lambda: codec.decode(iout, out=out)):
        time.sleep(0.2)
        for rep in range(4):   # the first call after the pause, then three back to back (sustained)
            t0 = time.perf_counter()
            f()
            print(f"call wall {1e3 * (time.perf_counter() - t0):.1f} ms")


def show(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    ev = [("K", s, e, n) for n, s, e in db.execute("select name, start, end from kernels")]
    mc = next((t for t in tabs if t == "memory_copies"), None)
    if mc:
        cols = [r[1] for r in db.execute(f"pragma table_info({mc})")]
        nm = "name" if "name" in cols else cols[0]
        sz = "size" if "size" in cols else "0"
        ev += [("C", s, e, f"{n} {b}") for n, s, e, b in db.execute(f"select {nm}, start, end, {sz} from {mc}")]
    else:
        print("no memory_copies view; tables:", tabs)
    ev.sort(key=lambda r: r[1])
    groups, cur = [], []
    for r in ev:
        if cur and r[1] - max(x[2] for x in cur) > 100e6:
            groups.append(cur)
            cur = []
        cur.append(r)
    groups.append(cur)
    for g in groups[-2:]:
        t0 = g[0][1]
        print(f"--- call: {len(g)} events, span {1e-6 * (max(x[2] for x in g) - t0):.2f} ms")
        # merge consecutive kernels into one compute interval
        i = 0
        while i < len(g):
            kind, s, e, n = g[i]
            if kind == "K":
                j, busy = i, 0
                while j < len(g) and g[j][0] == "K":
                    busy += g[j][2] - g[j][1]
                    e = max(e, g[j][2])
                    j += 1
                print(f"  +{1e-6 * (s - t0):8.2f} ms  kernels x{j - i:3d}  span {1e-6 * (e - s):7.2f} ms  busy {1e-6 * busy:7.2f} ms")
                i = j
            else:
                if e - s > 2e5:
                    print(f"  +{1e-6 * (s - t0):8.2f} ms  copy {n:>40s}  {1e-6 * (e - s):7.2f} ms")
                i += 1


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
