#!/bin/bash
# GPU box: HBM byte counters (separate --pmc passes, one group each) of the cyclic phase of one rank of the transposed pipeline, beside the
# same counters of the whole-grid passes.   tools/pmc_cyclic.sh r06 2048 8 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
round=${1:-r06}; n=${2:-2048}; world=${3:-8}; rank=${4:-3}
O=$R/gpurun_out/$round; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pmc() {  # name, counters...
  local name=$1; shift
  timeout -k 5 400 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${name}_cyclic_n$n -- python3 $R/tools/run_cyclic_rank.py $n $world $rank 1 > $O/pmc_${name}_cyclic_n$n.log 2>&1
  python3 $R/tools/pmc_summary.py $O/pmc_${name}_cyclic_n$n > $O/pmc_${name}_cyclic_n$n.summary.txt 2>&1
  rm -rf $O/pmc_${name}_cyclic_n$n $O/pmc_${name}_cyclic_n$n.log
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc l2 TCC_HIT_sum TCC_MISS_sum
ls $O | grep cyclic
