#!/bin/bash
# GPU box: per-kernel average durations of one bench run (rocprofv3 --kernel-trace --stats), printed as a table.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/kstats; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $O/log.txt 2>&1
python3 - "$O" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::|vp::", "", r["Name"])[:80]
    print("%-82s %4s avg %9.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
