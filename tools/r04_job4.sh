#!/bin/bash
# GPU box, round 4: compact id state (IdC) -- parity at n = 1152 / 1280, then n = 2048 bench against the 8-byte ids
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r04e; mkdir -p $O
export VPHIP_LIB=$R/tools/exp/libvphip_cpt2.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "compact_id_state" > $O/pytest_compact.txt 2>&1
tail -5 $O/pytest_compact.txt
timeout 900 python bench.py --grid-n 2048 --steps 4 --warmup 1 --no-cpu-baseline > $O/n2048_cpt.json 2> $O/n2048_cpt.err
ls -la $O
