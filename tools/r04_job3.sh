#!/bin/bash
# GPU box, round 4, third batch: copy-kernel shapes; nt stores + XCD-contiguous tile ranges (time and fabric bytes); n = 2048 variants.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04c; mkdir -p $O
timeout 300 tools/exp/copy > $O/copy_shapes.txt 2>&1
libs=tools/exp/libvphip_base.so,tools/exp/libvphip_nt2.so,tools/exp/libvphip_ntx1.so,tools/exp/libvphip_ntx2.so
timeout 600 python tools/ab_step.py --n 512 --libs $libs > $O/ab_xcd_512.txt 2>&1
timeout 900 python tools/ab_step.py --n 1024 --rounds 5 --libs $libs > $O/ab_xcd_1024.txt 2>&1
for v in ntx1 ntx2; do
  for n in 512 1024; do
    echo "==== $v n=$n FETCH_SIZE" >> $O/pmc_bytes_xcd.txt
    tools/pmc_grp.sh tools/exp/libvphip_$v.so $n "FETCH_SIZE" >> $O/pmc_bytes_xcd.txt 2>&1
  done
done
timeout 1200 python -m pytest tests/test_multi_gpu.py tests/test_cli.py -x -q -m gpu -k "csg_checks or gpus_flag" > $O/pytest_new.txt 2>&1
# n = 2048: the id passes on the round-1 tile kernel (default) / with nt stores / on the f64-pair tile kernel with nt stores
VPHIP_LIB=$R/tools/exp/libvphip_base.so timeout 900 python bench.py --grid-n 2048 --steps 4 --warmup 1 --no-cpu-baseline > $O/n2048_base.json 2> $O/n2048_base.err
VPHIP_LIB=$R/tools/exp/libvphip_znt.so timeout 900 python bench.py --grid-n 2048 --steps 4 --warmup 1 --no-cpu-baseline > $O/n2048_znt.json 2> $O/n2048_znt.err
VP_JFA_DENSE_WIDE=1 VPHIP_LIB=$R/tools/exp/libvphip_znt.so timeout 900 python bench.py --grid-n 2048 --steps 4 --warmup 1 --no-cpu-baseline > $O/n2048_znt_dense.json 2> $O/n2048_znt_dense.err
ls -la $O
