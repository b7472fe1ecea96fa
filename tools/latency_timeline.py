#!/usr/bin/env python3
"""Timeline of the kernels of ONE small-batch encode and ONE decode call, from a rocprofv3 kernel trace.

    cd /tmp && rocprofv3 --kernel-trace -d $OUT/lat -o lat -- python tools/latency_timeline.py run 64
    python tools/latency_timeline.py show $OUT/lat/lat_results.db

'run' issues warm-ups, then a marker gap, then the measured calls; 'show' prints, for the LAST encode and the LAST decode
call in the trace, every kernel's start offset, duration and the idle gap to its predecessor.
"""
import os
import re
import sqlite3
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(n):
    import torch
    from vqvdb_amd import synth, weightpack
    from vqvdb_amd.codec import HipCodec
    codec = HipCodec(weightpack.dumps(synth.make_weights(0)))
    x = torch.rand(n, 512, device="cuda")
    idx = torch.empty(n, 64, dtype=torch.uint8, device="cuda")
    rec = torch.empty(n, 512, device="cuda")
    for _ in range(5):
        codec.encode_device(x.data_ptr(), n, idx.data_ptr())
        codec.decode_device(idx.data_ptr(), n, rec.data_ptr())
    torch.cuda.synchronize()
    for leg in (lambda: codec.encode_device(x.data_ptr(), n, idx.data_ptr()), lambda: codec.decode_device(idx.data_ptr(), n, rec.data_ptr())):
        time.sleep(0.05)
        t0 = time.perf_counter()
        leg()
        torch.cuda.synchronize()
        print(f"call wall {1e3 * (time.perf_counter() - t0):.3f} ms")


def show(path):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start"))
    # calls are separated by >= 20 ms of idle time
    groups, cur = [], []
    for r in rows:
        if cur and r[1] - cur[-1][2] > 20e6:
            groups.append(cur)
            cur = []
        cur.append(r)
    groups.append(cur)
    for g in groups[-2:]:
        t0 = g[0][1]
        print(f"--- {len(g)} kernels, first start -> last end {1e-3 * (g[-1][2] - t0):.1f} us, sum of durations {1e-3 * sum(r[2] - r[1] for r in g):.1f} us")
        prev = t0
        for name, s, e, gx, gy, wg in g:
            k = re.sub(r"\(.*", "", name).replace("void ", "")[:70]
            print(f"  +{1e-3 * (s - t0):8.1f} us  dur {1e-3 * (e - s):7.1f}  gap {1e-3 * (s - prev):6.1f}  grid {gx // wg}x{gy} wg {wg}  {k}")
            prev = e


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        show(sys.argv[2])
