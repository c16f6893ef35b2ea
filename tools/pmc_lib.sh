#!/bin/bash
# dev helper (GPU box): SQ counter groups for one experimental library:  tools/pmc_lib.sh tools/exp/libvphip_X.so [n]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
lib=$1; n=${2:-512}
name=$(basename $lib .so)
export VPHIP_LIB=$R/$lib
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_${name}_$i -- python3 $R/tools/run_passes.py $n 1 > $R/gpurun_out/pmc_${name}_$i.log 2>&1
  echo "---- $name group $i"
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${name}_$i | grep -A9 "dense\|zstream"
  rm -rf $R/gpurun_out/pmc_${name}_$i
done
