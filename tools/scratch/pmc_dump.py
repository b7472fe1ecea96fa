import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, grid_size_x, counter_name, avg(value), avg(end-start) from counters_collection group by kernel_name, grid_size_x, counter_name").fetchall()
d = {}
for k, gx, c, v, dur in rows:
    k = re.sub(r"\(.*", "", k).replace("void ", "")[:60]
    if "at::" in k or "rocclr" in k or dur < 200e3: continue
    d.setdefault((k, gx), {})[c] = v
    d[(k, gx)]["dur_us"] = dur / 1e3
for (k, gx), v in sorted(d.items(), key=lambda kv: -kv[1]["dur_us"]):
    wc = max(v.get("SQ_WAVE_CYCLES", 1), 1)
    print(k, gx, f"{v['dur_us']:.0f}us", " ".join(f"{c[3:]}={x / wc:.3f}" for c, x in v.items() if c.startswith("SQ_") and c != "SQ_WAVE_CYCLES"))
