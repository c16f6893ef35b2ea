#!/bin/bash
# GPU box, round 4, second batch: formulation floor, ablations of the dense tile kernel (time AND fabric bytes), new tests, bench line.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04b; mkdir -p $O
timeout 300 tools/exp/floor 1024 > $O/floor_n1024.txt 2>&1
timeout 300 tools/exp/floor 512 > $O/floor_n512.txt 2>&1
libs=tools/exp/libvphip_base.so,tools/exp/libvphip_nogather.so,tools/exp/libvphip_hot.so,tools/exp/libvphip_nolds.so,tools/exp/libvphip_hotnolds.so
timeout 600 python tools/ab_step.py --n 512 --libs $libs > $O/ab_ablate_512.txt 2>&1
timeout 900 python tools/ab_step.py --n 1024 --rounds 5 --libs $libs > $O/ab_ablate_1024.txt 2>&1
for v in base nt sc1 nogather; do
  for n in 512 1024; do
    for c in FETCH_SIZE WRITE_SIZE; do
      echo "==== $v n=$n $c" >> $O/pmc_bytes.txt
      tools/pmc_grp.sh tools/exp/libvphip_$v.so $n "$c" >> $O/pmc_bytes.txt 2>&1
    done
  done
done
timeout 1500 python -m pytest tests/test_multi_gpu.py tests/test_gpu_parity.py tests/test_slab_gpu.py -x -q -m gpu -k "unproduced or csg_checks or jfa_run_needs or stream_copy or shared_gpu" > $O/pytest_new.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
ls -la $O
