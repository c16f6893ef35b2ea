#!/bin/bash
# GPU box: the evidence of a round in one job -- per-round profiles (tools/profile_round.sh), the slab-scaling tables, the randomized
# parity campaigns and the size sweep.   tools/round_job.sh r05 [fuzz seconds]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
round=${1:-r05}; secs=${2:-240}
O=$R/gpurun_out/$round; mkdir -p $O
bash tools/profile_round.sh $round > $O/profile_round.log 2>&1
cd $R
for n in 512 1024 2048; do timeout 900 python tools/slab_scaling.py $n 2>&1 | grep -v amdgpu.ids > $O/slab_scaling_n$n.txt; done
timeout $((secs + 120)) python tools/fuzz_pipelines.py --seconds $secs 2>&1 | grep -v amdgpu.ids | tail -400 > $O/fuzz_pipelines.txt
timeout $((secs + 120)) python tools/fuzz_slabs.py --seconds $secs 2>&1 | grep -v amdgpu.ids | tail -400 > $O/fuzz_slabs.txt
timeout $((secs + 120)) python tools/fuzz_parity.py --seconds $secs 2>&1 | grep -v amdgpu.ids | tail -400 > $O/fuzz_parity.txt
bash tools/size_sweep.sh 2>&1 | grep -v amdgpu.ids > $O/size_sweep.txt
tail -2 $O/fuzz_*.txt; ls $O
