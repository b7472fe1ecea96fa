#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite) outputs into a small text table for profiles/.

    python tools/rocpd_summary.py --kt gpurun_out/prof/kt/r01_results.db \
        [--fetch .../pmc_fetch/r01_results.db] [--write .../pmc_write/r01_results.db] > profiles/rNN_xxx.txt

Per kernel: calls, average duration (kernel-trace), and HBM traffic per launch from the PMC
passes.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reads exactly 1/2 of the bytes
of wide (16 B/lane) coalesced reads -> 'fetch_MB_x2' doubles it; WRITE_SIZE is uncalibrated.
"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return name if len(name) < 90 else name[:87] + "..."


def pmc(path, counter):
    """(kernel, grid) -> (launches, average counter value)"""
    db = sqlite3.connect(path)
    out = {}
    for name, gx, gy, n, avg in db.execute(
            "select kernel_name, grid_size_x, grid_size_y, count(*), avg(value) from counters_collection where counter_name=? "
            "group by kernel_name, grid_size_x, grid_size_y", (counter,)):
        out[(short(name), gx, gy)] = (n, avg)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kt", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    a = ap.parse_args()
    db = sqlite3.connect(a.kt)
    # one row per (kernel, launch geometry): the same kernel is launched at 65536 leaves by the throughput legs and at small
    # batches (or with gridDim.y > 1, the position-split path) by the training leg; averaging across them would be meaningless
    rows = list(db.execute("select name, count(*), avg(duration), sum(duration), max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(scratch_size),"
                           " max(workgroup_x), grid_x, grid_y from kernels group by name, grid_x, grid_y order by sum(duration) desc"))
    f = pmc(a.fetch, "FETCH_SIZE") if a.fetch else {}
    w = pmc(a.write, "WRITE_SIZE") if a.write else {}
    tot = sum(r[3] for r in rows)
    print(f"{'kernel':90s} {'calls':>5s} {'avg_us':>10s} {'%':>6s} {'vgpr':>5s} {'agpr':>5s} {'lds':>7s} {'scr':>4s} {'wg':>4s} {'grid':>8s} {'fetch_MB':>9s} {'fetch_MB_x2':>11s} {'write_MB':>9s}")
    for name, n, avg, s, vg, ag, lds, scr, wg, gx, gy in rows:
        k = short(name)
        grid = gx * gy
        if "at::native" in k or "rocclr" in k:
            continue
        fm = f.get((k, gx, gy), (0, None))[1]
        wm = w.get((k, gx, gy), (0, None))[1]
        print(f"{k:90s} {n:5d} {avg / 1e3:10.1f} {100 * s / tot:6.2f} {vg:5d} {ag:5d} {lds:7d} {scr:4d} {wg:4d} {grid:8d} "
              f"{(fm / 1024 if fm is not None else float('nan')):9.1f} {(fm / 512 if fm is not None else float('nan')):11.1f} {(wm / 1024 if wm is not None else float('nan')):9.1f}")


if __name__ == "__main__":
    main()
