"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel name."""
import sys, csv, glob, collections, os, re
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.search(r"(vp::)?([A-Za-z_0-9]+(<[^>]*>)?)\(", name)
    return m.group(2) if m else name[:50]
for f in files:
    for row in csv.DictReader(open(f)):
        acc[short(row.get("Kernel_Name", "?"))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%3d mean=%.5g" % (c, len(v), sum(v) / len(v)))
