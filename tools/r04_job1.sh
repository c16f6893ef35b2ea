#!/bin/bash
# GPU box, round 4, first batch: (1) does straight-line code length cost issue rate (tools/ubench/icache.hip), (2) instruction-cache
# counters of the dense tile kernels, (3) cache policy of the tile kernel's output stores: time AND fabric bytes per variant.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04a; mkdir -p $O
timeout 300 tools/exp/icache > $O/icache.txt 2>&1
libs=tools/exp/libvphip_base.so,tools/exp/libvphip_nt.so,tools/exp/libvphip_sc1.so,tools/exp/libvphip_sc01.so
timeout 600 python tools/ab_step.py --n 512 --libs $libs > $O/ab_store_512.txt 2>&1
timeout 900 python tools/ab_step.py --n 1024 --rounds 5 --libs $libs > $O/ab_store_1024.txt 2>&1
for v in base nt sc1 sc01; do
  for n in 512 1024; do
    echo "==== $v n=$n FETCH_SIZE" >> $O/pmc_store.txt
    tools/pmc_grp.sh tools/exp/libvphip_$v.so $n "FETCH_SIZE" >> $O/pmc_store.txt 2>&1
    echo "==== $v n=$n WRITE_SIZE" >> $O/pmc_store.txt
    tools/pmc_grp.sh tools/exp/libvphip_$v.so $n "WRITE_SIZE" >> $O/pmc_store.txt 2>&1
  done
done
for n in 512 1024; do
  echo "==== base n=$n icache" >> $O/pmc_icache.txt
  tools/pmc_grp.sh tools/exp/libvphip_base.so $n "SQ_IFETCH SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "dense\|zstream\|first_two" >> $O/pmc_icache.txt 2>&1
  tools/pmc_grp.sh tools/exp/libvphip_base.so $n "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES" "dense\|zstream\|first_two" >> $O/pmc_icache.txt 2>&1
done
ls -la $O
