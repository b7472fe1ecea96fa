#!/usr/bin/env python3
"""L2 (TCC) view of every kernel of the bench from a rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum pass (rocpd
sqlite) next to the FETCH_SIZE pass: how many requests reach the L2, how many it answers itself (weights of a layer are <= 1.8 MB and
stay resident; activations re-read within the reuse distance the 4 MB of an XCD's L2 covers), and how many go on to the fabric
(Infinity Cache / HBM) = what FETCH_SIZE counts.  One request = one 128-byte line (gfx950).  VERDICT r2 item 6: attribute the re-fetch of
the 4^3 convs to weights vs activations.

    python tools/pmc_tcc_summary.py <tcc_pass.db> [<fetch_pass.db>]
"""
import re
import sqlite3
import sys


def table(path):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, grid_size_x, grid_size_y, counter_name, avg(value) from counters_collection "
                      "group by kernel_name, grid_size_x, grid_size_y, counter_name").fetchall()
    d, best = {}, {}
    for k, gx, gy, c, v in rows:   # per kernel keep the unsplit launches with the largest grid (the 65536-leaf ones)
        k = re.sub(r"\(.*", "", k).replace("void ", "")
        if "at::" in k or "rocclr" in k or gy > 2 or gx < best.get(k, 0):
            continue
        if gx > best.get(k, 0):
            best[k] = gx
            d[k] = {}
        d[k][c] = v
    return d


def main():
    t = table(sys.argv[1])
    f = table(sys.argv[2]) if len(sys.argv) > 2 else {}
    print(f"{'kernel':78s} {'L2 req GB':>10s} {'reads GB':>9s} {'hit rate':>9s} {'miss GB':>8s} {'FETCH_SIZE x2 GB':>17s}")
    for k, v in sorted(t.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", 0)):
        hit, miss, req, rd = (v.get(n, 0.0) for n in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum"))
        if req < 1e4:
            continue
        fe = f.get(k, {}).get("FETCH_SIZE")
        print(f"{k[:78]:78s} {req * 128 / 1e9:10.2f} {rd * 128 / 1e9:9.2f} {hit / max(hit + miss, 1):9.3f} {miss * 128 / 1e9:8.2f} "
              f"{(fe * 2048 / 1e9 if fe is not None else float('nan')):17.2f}")


if __name__ == "__main__":
    main()
