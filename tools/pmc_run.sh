#!/bin/bash
# dev helper (GPU box): collect PMC counter groups over tools/run_passes.py, one rocprofv3 pass per group, and summarise.
# Every pass runs under `timeout`: a counter group the hardware cannot schedule makes rocprofv3 abort and then hang.
# usage: tools/pmc_run.sh OUTNAME "CTR1 CTR2 ..." ["CTR..." ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
name=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_${name}_$i -- python3 $R/tools/run_passes.py 512 1 > $R/gpurun_out/pmc_${name}_$i.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${name}_$i > $R/gpurun_out/pmc_${name}_$i.summary.txt 2>&1
done
