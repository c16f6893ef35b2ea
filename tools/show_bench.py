"""Dev: pretty-print a bench.py JSON line (file argument)."""
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], d["unit"], " ms/step", d["ms_per_step"], " n_gpus", d["n_gpus"])
print("roofline", d["roofline"])
for k, v in d["kernels"].items():
    print("  %-12s %s" % (k, v))
if "n1024" in d:
    print("n1024", {k: v for k, v in d["n1024"].items() if k != "kernels"})
    for k, v in d["n1024"]["kernels"].items():
        print("  1024 %-12s %s" % (k, v))
print("cpu_baseline", d.get("cpu_baseline"))
print("multi", d.get("multi"))
