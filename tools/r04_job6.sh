#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04f; mkdir -p $O
export VPHIP_LIB=$R/tools/exp/libvphip_cur.so
timeout 2400 python -m pytest tests/test_multi_gpu.py -x -q -m gpu -k "matches_single_and_oracle or hybrid or headline_size or unproduced" > $O/pytest_multi.txt 2>&1
tail -5 $O/pytest_multi.txt
VPHIP_LIB=$R/tools/exp/libvphip_cptpipe.so timeout 900 python bench.py --grid-n 2048 --steps 4 --warmup 1 --no-cpu-baseline > $O/n2048_pipe.json 2> $O/n2048_pipe.err
VPHIP_LIB=$R/tools/exp/libvphip_cur.so timeout 900 python bench.py --grid-n 2048 --steps 4 --warmup 1 --no-cpu-baseline > $O/n2048_cur.json 2> $O/n2048_cur.err
ls -la $O
