"""Dev tool: rocprofv3 --pmc CSV -> one line per DISPATCH of the kernels whose name matches (dispatch order = pass order)."""
import sys, csv, glob, collections, os
d, pat = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r.get("Kernel_Name", ""):
            key = int(r["Dispatch_Id"])
            rows.setdefault(key, {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(rows):
    r = rows[k]
    tpl = r["name"].split("jfa_pass_dense<")[-1].split(">(")[0].replace("vp::(anonymous namespace)::", "")[:60]
    print(k, tpl, " ".join("%s=%.4g" % (c, v) for c, v in r.items() if c != "name"))
