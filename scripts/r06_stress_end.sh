#!/bin/bash
# Round 6, end of round: the randomised parity sweeps on the final kernels (block kernel in both shapes).
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_stress_end
mkdir -p "$OUT"
cd "$R"
timeout 900 python scripts/experiments/stress_observable_blocks.py 48 100 > "$OUT/blocks_bb13.txt" 2>&1; tail -2 "$OUT/blocks_bb13.txt"
QHBM_OBS_BLOCK_BITS=12 timeout 900 python scripts/experiments/stress_observable_blocks.py 48 100 > "$OUT/blocks_bb12.txt" 2>&1; tail -2 "$OUT/blocks_bb12.txt"
timeout 900 python scripts/experiments/stress_default_plans.py 30 500 > "$OUT/default_plans.txt" 2>&1; tail -2 "$OUT/default_plans.txt"
timeout 900 python scripts/experiments/stress_api_sizes.py > "$OUT/api_sizes.txt" 2>&1; tail -2 "$OUT/api_sizes.txt"
