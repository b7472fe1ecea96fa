#!/usr/bin/env python3
"""Every counter of a rocprofv3 --pmc pass per kernel (rocpd sqlite): average value per launch, and as a fraction of SQ_WAVE_CYCLES
where that counter was collected in the same pass.  usage: pmc_any_summary.py results.db [kernel-name-substring]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select kernel_name, counter_name, avg(value), avg(end-start), count(*) from counters_collection "
                  "group by kernel_name, counter_name").fetchall()
d = {}
for k, c, v, dur, n in rows:
    k = re.sub(r"\(.*", "", k).replace("void ", "")
    if flt not in k or "at::" in k:
        continue
    d.setdefault(k, {"dur": dur, "n": n})[c] = v
for k, v in sorted(d.items(), key=lambda kv: -kv[1]["dur"]):
    wc = v.get("SQ_WAVE_CYCLES")
    print(f"{k[:100]}  avg {v['dur'] / 1e3:.1f} us  ({v['n']} launches)")
    for c, x in sorted(v.items()):
        if c in ("dur", "n"):
            continue
        print(f"    {c:32s} {x:16.0f}" + (f"   {x / wc:8.3f} of SQ_WAVE_CYCLES" if wc and c != "SQ_WAVE_CYCLES" else ""))
