"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel name."""
import sys, csv, glob, collections, os, re
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
def short(name):
    """kernel name without return type, namespaces and the argument list (template arguments kept, nesting respected)"""
    name = name.replace("(anonymous namespace)::", "").replace("vp::", "")
    if name.startswith("void "):
        name = name[5:]
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name[:80]
for f in files:
    for row in csv.DictReader(open(f)):
        acc[short(row.get("Kernel_Name", "?"))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%3d mean=%.5g" % (c, len(v), sum(v) / len(v)))
