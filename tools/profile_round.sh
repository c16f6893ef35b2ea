#!/bin/bash
# GPU box: produce the per-round evidence under gpurun_out/<round>/ (copy what should be judged into profiles/<round>/).
#   tools/profile_round.sh r01
# Every rocprofv3 pass runs under `timeout`: a counter group the hardware cannot schedule makes rocprofv3 abort and hang.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
round=${1:-r01}
O=$R/gpurun_out/$round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep '^{"metric"' $O/bench_under_rocprof.log > $O/bench_under_rocprof.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv 2>/dev/null
pmc() {  # name, counters...
  local name=$1; shift
  timeout -k 5 200 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -- python3 $R/tools/run_passes.py 512 1 > $O/pmc_$name.log 2>&1
  python3 $R/tools/pmc_summary.py $O/pmc_$name > $O/pmc_$name.summary.txt 2>&1
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc l2 TCC_HIT_sum TCC_MISS_sum
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
pmc sq2 GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pmc tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum
pmc td TD_TD_BUSY_sum TD_TC_STALL_sum
rm -rf $O/stats $O/pmc_*/ $O/*.log
ls $O
