#!/bin/bash
# GPU box: the whole GPU suite, smoke, and the three bench sizes with the in-tree library
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/${1:-r04full}; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
timeout 600 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 python bench.py --grid-n 1024 --no-cpu-baseline > $O/bench_n1024.json 2> $O/bench_n1024.err
timeout 900 python bench.py --grid-n 2048 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_n2048.json 2> $O/bench_n2048.err
ls -la $O
