#!/bin/bash
# dev helper (GPU box): one counter group for one library:  tools/pmc_grp.sh LIB N "CTR CTR ..."
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
lib=$1; n=$2; grp=$3; pat=${4:-dense|zstream}
name=$(basename $lib .so)
[ -f "$R/$lib" ] && export VPHIP_LIB=$R/$lib
cd /tmp && export TMPDIR=/tmp
timeout -k 5 150 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmcg_$name -- python3 $R/tools/run_passes.py $n 1 > $R/gpurun_out/pmcg_$name.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcg_$name | grep -E -A10 "$pat"
rm -rf $R/gpurun_out/pmcg_$name
