#!/usr/bin/env python3
"""profiles/hbm_traffic_per_launch.json from the FETCH_SIZE / WRITE_SIZE PMC passes (rocpd sqlite).
bench.py reports roofline.traffic from this file (it cannot collect PMC counters itself).
FETCH_SIZE is doubled per the gfx950 calibration of MI355X_MICROARCH.md §HBM (verified here: the
elementwise kernel's corrected 2050 MB vs 2048 MB algorithmic); WRITE_SIZE is used as reported."""
import hashlib
import json
import os
import re
import sqlite3
import sys


def kernel_build_id():
    """sha256 (12 hex) over the kernel sources of vqvdb_amd/csrc, in name order: what the counters were collected on.  bench.py computes
    the same id from the tree it runs and flags roofline.traffic as stale when they differ."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vqvdb_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(root)):
        if name.endswith((".hip", ".h", ".inc")):
            h.update(name.encode())
            h.update(open(os.path.join(root, name), "rb").read())
    return h.hexdigest()[:12]

LAUNCH = [  # (regex on the kernel's demangled name, launch name used by the library's profiler)
    (r"conv_first_k<0[,>]", "enc_conv_first_stats"), (r"conv_first_k<1[,>]", "enc_conv_first_gn"),
    (r"conv_first_roll_k<0[,>]", "enc_conv_first_stats"), (r"conv_first_roll_k<1[,>]", "enc_conv_first_gn"),   # round 4: rolling row window (normalising pass)
    (r"conv8_c16_k<4, false, true, false>", "enc_res16_conv1"), (r"conv8_c16_k<4, true, false, false>", "enc_res16_conv2"),
    (r"conv8_lds_k<false, true", "enc_res16_conv1"), (r"conv8_lds_k<true, false", "enc_res16_conv2"),   # persistent grid = CUs: the large-batch launches
    (r"conv_rows16_k<16, 32, 8, 4, 4, 2, 1, 0, false, 8, false, true[,>]", "enc_down"),
    (r"conv_rows16_k<32, 32, 4, 4, 3, 1, 1, 1, false, 8, false, true[,>]", "enc_res32_conv1"),
    (r"conv_rows16_k<32, 32, 4, 4, 3, 1, 1, 1, true, 0, true, true[,>]", "enc_res32_conv2"),
    (r"conv_down_lds_k", "enc_down"), (r"conv4_lds_k<false, true, false", "enc_res32_conv1"), (r"conv4_lds_k<true, false, true", "enc_res32_conv2"),   # round 4: LDS plane rings
    (r"vq_folded_k<8[,>]", "enc_vq"), (r"pack_leaves_k", "pack_leaves"), (r"stem_lut_k", "dec_stem"),
    (r"gn_relu_stats_k<64", "dec_gn_relu_stats"), (r"stem_fused_k", "dec_stem_gn"), (r"stem_taps_k", "dec_stem_gn"),
    (r"conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, false, 8, false, false[,>]", "dec_res64_conv1"),
    (r"conv_rows16_k<64, 64, 4, 4, 3, 1, 1, 1, true, 0, true, false[,>]", "dec_res64_conv2"),
    (r"tail_rows16_k<", "dec_tail"), (r"tail_rows32_k<", "dec_tail_rows32"), (r"tail_groups16_k<", "dec_tail_groups"), (r"conv_mfma32_k<64, 128, 64, 4, 8,", "dec_tail_slab"),
]


def read(path, counter):
    """Per launch name: the LARGEST counter value over the launches of the kernels that map to it = the 65536-leaf launches of the
    throughput legs (the same kernels also run at small batches and with gridDim.y > 1; a persistent kernel has the same grid for its
    65536-leaf launches and for mid-size passes; two kernels may share a launch name: stem_taps_k for full chunks, stem_fused_k below)."""
    db = sqlite3.connect(path)
    out = {}
    for name, grid, gy, top in db.execute("select kernel_name, grid_size_x, grid_size_y, max(value) from counters_collection where counter_name=? "
                                          "group by kernel_name, grid_size_x, grid_size_y", (counter,)):
        if gy != 1 and not (gy == 2 and re.search(r"vq_folded_k<8[,>]", name)):   # position-split launches (small batches); the full-chunk VQ search runs two position ranges per tile
            continue
        for rx, launch in LAUNCH:
            if re.search(rx, name) and not name.startswith("build_"):
                out[launch] = max(out.get(launch, 0.0), top * 1024.0)
    return out


def main():
    fetch_db, write_db, source, out = sys.argv[1:5]
    f, w = read(fetch_db, "FETCH_SIZE"), read(write_db, "WRITE_SIZE")
    build = kernel_build_id()
    res = {k: {"fetch_bytes": round(2.0 * f[k]), "write_bytes": round(w.get(k, 0.0)), "leaves_per_launch": 65536, "source": source, "build": build,
               "correction": "FETCH_SIZE x2 (gfx950, wide coalesced reads), WRITE_SIZE as reported"} for k in f}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(out, len(res), "kernels")


if __name__ == "__main__":
    main()
