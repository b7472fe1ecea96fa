#!/usr/bin/env python3
"""Short device-resident bench of the two passes, per-kernel table on stdout (for A/B runs of kernel variants selected by environment
variables on the GPU box):   VQHIP_FIRST_SRC=packed python tools/bench_kernels.py [steps]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "24"
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "2", "--no-train", "--no-host-path", "--no-cpu-baseline"],
                   capture_output=True, text=True)
if r.returncode != 0:
    sys.stderr.write(r.stderr[-2000:])
    sys.exit(1)
d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("VQHIP_")) or "default"
print(f"[{tag}] encode {d['value'] / 1e6:.4f} M leaves/s ({d['ms_per_step']:.4f} ms)  decode {d['decode_value'] / 1e6:.4f} M ({d['decode_ms_per_step']:.4f} ms)")
for leg in ("encode", "decode"):
    print("   " + "  ".join(f"{k['kernel']} {k['avg_ms']:.4f}" for k in d["kernels"][leg]))
sm = d.get("small_batch") or {}
print("   small: " + "  ".join(f"{k}: enc {v['encode_ms']:.4f} dec {v['decode_ms']:.4f}" for k, v in sm.items() if isinstance(v, dict)))
