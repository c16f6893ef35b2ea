#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04l; mkdir -p $O
for n in 512 1024; do
  echo "==== n=$n utcl1" >> $O/pmc_tlb.txt
  tools/pmc_grp.sh none $n "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_PERMISSION_MISS_sum" "dense|first_two|vox_fill" >> $O/pmc_tlb.txt 2>&1
  echo "==== n=$n stalls" >> $O/pmc_tlb.txt
  tools/pmc_grp.sh none $n "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" "dense|first_two|vox_fill" >> $O/pmc_tlb.txt 2>&1
  echo "==== n=$n latency" >> $O/pmc_tlb.txt
  tools/pmc_grp.sh none $n "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" "dense|first_two|vox_fill" >> $O/pmc_tlb.txt 2>&1
done
cat $O/pmc_tlb.txt | grep -v "^--$"
