#!/bin/bash
# GPU box: interleaved A/B of the whole step per kernel for a list of experimental libraries, n = 512 and 1024 (+ optional n = 2048 bench)
#   tools/ab_job.sh OUTNAME name1,name2,... [2048]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/$1; mkdir -p $O
libs=$(echo $2 | tr ',' '\n' | sed "s#^#tools/exp/libvphip_#; s#\$#.so#" | paste -sd,)
timeout 600 python tools/ab_step.py --n 512 --libs $libs 2>&1 | grep -v amdgpu.ids > $O/ab_512.txt
timeout 900 python tools/ab_step.py --n 1024 --rounds 5 --libs $libs 2>&1 | grep -v amdgpu.ids > $O/ab_1024.txt
if [ "$3" = "2048" ]; then
  for v in $(echo $2 | tr ',' ' '); do
    VPHIP_LIB=$R/tools/exp/libvphip_$v.so timeout 900 python bench.py --grid-n 2048 --steps 4 --warmup 1 --no-cpu-baseline > $O/n2048_$v.json 2> $O/n2048_$v.err
  done
fi
cat $O/ab_512.txt $O/ab_1024.txt
