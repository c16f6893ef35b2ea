#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04m; mkdir -p $O
timeout 3000 python -m pytest tests/test_slab_gpu.py tests/test_multi_gpu.py -x -q -m gpu > $O/pytest_slab_multi.txt 2>&1
tail -15 $O/pytest_slab_multi.txt
