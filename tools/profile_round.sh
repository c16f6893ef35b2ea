#!/bin/bash
# GPU box: produce the per-round evidence under gpurun_out/<round>/ (copy what should be judged into profiles/<round>/).
#   tools/profile_round.sh r02
# Every rocprofv3 pass runs under `timeout`: a counter group the hardware cannot schedule makes rocprofv3 abort and hang.
# PMC passes are separate runs (never combined with tracing), one counter group each.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
round=${1:-r02}
O=$R/gpurun_out/$round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace of the bench command itself (n = 512 headline + its n = 1024 block), then the same command un-profiled
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep '^{"metric"' $O/bench_under_rocprof.log > $O/bench_under_rocprof.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv 2>/dev/null
python3 $R/bench.py > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
pmc() {  # size, name, counters...
  local n=$1 name=$2; shift 2
  timeout -k 5 250 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${name}_n$n -- python3 $R/tools/run_passes.py $n 1 > $O/pmc_${name}_n$n.log 2>&1
  python3 $R/tools/pmc_summary.py $O/pmc_${name}_n$n > $O/pmc_${name}_n$n.summary.txt 2>&1
}
for n in 512 1024; do
  pmc $n fetch FETCH_SIZE
  pmc $n write WRITE_SIZE
  pmc $n l2 TCC_HIT_sum TCC_MISS_sum
  pmc $n sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
  pmc $n sq2 GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
done
pmc 512 tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum
python3 $R/tools/traffic_json.py $O > $O/jfa_dense_traffic.json 2> $O/traffic_json.err
rm -rf $O/stats $O/pmc_*/ $O/*.log
ls $O
# round 4: the other two sizes, the n = 2048 byte counters, the probes
python3 $R/bench.py --grid-n 1024 --no-cpu-baseline > $O/bench_n1024.json 2> $O/bench_n1024.err
python3 $R/bench.py --grid-n 2048 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_n2048.json 2> $O/bench_n2048.err
pmc 2048 fetch FETCH_SIZE
pmc 2048 write WRITE_SIZE
pmc 2048 sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
pmc 2048 sq2 GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
[ -x $R/tools/exp/floor ] || { mkdir -p $R/tools/exp; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $R/tools/ubench/floor.hip -o $R/tools/exp/floor; }
[ -x $R/tools/exp/floor ] && { $R/tools/exp/floor 1024 > $O/floor_n1024.txt 2>&1; $R/tools/exp/floor 512 > $O/floor_n512.txt 2>&1;
                               python3 $R/tools/floor_json.py $O ${round#r} > $O/formulation_floor.json; }
rm -rf $O/pmc_*/ $O/*.log
ls $O
