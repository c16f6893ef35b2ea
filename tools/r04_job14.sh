#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04n; mkdir -p $O
for n in 512 1024 2048; do
  timeout 1500 python tools/slab_scaling.py $n 2>&1 | grep -v amdgpu.ids > $O/slab_scaling_n$n.txt
done
VP_GHOST_VOLUME=0 timeout 1500 python tools/slab_scaling.py 2048 2>&1 | grep -v amdgpu.ids | head -8 > $O/slab_scaling_n2048_8byte_ids.txt
head -12 $O/slab_scaling_n512.txt $O/slab_scaling_n1024.txt $O/slab_scaling_n2048.txt $O/slab_scaling_n2048_8byte_ids.txt
