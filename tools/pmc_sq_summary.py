#!/usr/bin/env python3
"""MFMA utilisation / effective clock per kernel from a rocprofv3 --pmc pass (rocpd sqlite):
counters GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_WAIT_ANY.  GRBM_GUI_ACTIVE is summed over the 8 XCDs, MFMA busy over the
1024 SIMDs -> clock = GUI/8/duration, mfma_util = MFMA_BUSY / (GUI/8 * 1024)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, grid_size_x, grid_size_y, counter_name, avg(value), avg(end-start) from counters_collection "
                  "group by kernel_name, grid_size_x, grid_size_y, counter_name").fetchall()
d, best = {}, {}
for k, gx, gy, c, v, dur in rows:   # per kernel keep the unsplit launches with the largest grid (the 65536-leaf ones)
    k = re.sub(r"\(.*", "", k).replace("void ", "")
    if "at::" in k or "rocclr" in k or gy > 2 or gx < best.get(k, 0):   # (gy = 2: the VQ search of full chunks)
        continue
    if gx > best.get(k, 0):
        best[k] = gx
        d[k] = {}
    d[k][c] = v
    d[k]["dur"] = dur
print(f"{'kernel':78s} {'avg_us':>9s} {'clk_GHz':>8s} {'mfma_util':>9s} {'wait_inst':>9s} {'wait_any':>9s} {'active':>7s}")
for k, v in sorted(d.items(), key=lambda kv: -kv[1]["dur"]):
    g = v.get("GRBM_GUI_ACTIVE", 0) / 8.0
    wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
    print(f"{k[:78]:78s} {v['dur'] / 1e3:9.1f} {g / max(v['dur'], 1):8.2f} {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(g * 1024, 1):9.3f} "
          f"{v.get('SQ_WAIT_INST_ANY', 0) / wc:9.2f} {v.get('SQ_WAIT_ANY', 0) / wc:9.2f} {v.get('SQ_ACTIVE_INST_ANY', 0) / wc:7.2f}")
