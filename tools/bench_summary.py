"""Compact view of a bench.py JSON line: python tools/bench_summary.py bench.json"""
import json, sys
d = json.load(open(sys.argv[1]))
print("ENC %.3f M leaves/s  %.3f ms/step  whole-path frac %.3f" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["whole_path_frac"]))
for k in d["kernels"]["encode"]:
    print("   %-24s %8.4f ms  issued %6.1f TF  frac %.3f" % (k["kernel"], k["avg_ms"], k["tflops_issued"], k["tflops_issued"] / 157.3))
print("DEC %.3f M leaves/s  %.3f ms/step  whole-path frac %.3f" % (d["decode"]["value"] / 1e6, d["decode"]["ms_per_step"], d["decode"]["roofline"]["whole_path_frac"]))
for k in d["kernels"]["decode"]:
    print("   %-24s %8.4f ms  issued %6.1f TF  frac %.3f" % (k["kernel"], k["avg_ms"], k["tflops_issued"], k["tflops_issued"] / 157.3))
