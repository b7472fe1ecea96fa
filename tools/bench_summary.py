import json, sys
d=json.load(open(sys.argv[1]))
print("ENC", d["value"], d["ms_per_step"], d["roofline"]["whole_path_frac"])
for k in d["kernels"]["encode"]: print("  ", k)
print("DEC", d["decode"]["value"], d["decode"]["ms_per_step"], d["decode"]["roofline"]["whole_path_frac"])
for k in d["kernels"]["decode"]: print("  ", k)
