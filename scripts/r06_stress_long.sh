#!/bin/bash
# Round 6, end of round: long randomised parity sweeps on the final kernels with NEW seeds (unused GPU minutes).
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_stress_long
mkdir -p "$OUT"
cd "$R"
run() { local name=$1; shift; ( time timeout 1500 "$@" ) > "$OUT/$name.txt" 2>&1; echo "$name: $(grep -i 'done' "$OUT/$name.txt" | tail -1)"; }
run measure_15_20 python scripts/experiments/stress_measure.py 36 2000
STRESS_N_BASE=21 STRESS_N_SPAN=3 run measure_21_23 python scripts/experiments/stress_measure.py 6 2100
run default_plans python scripts/experiments/stress_default_plans.py 48 2000
run parity_tiles python scripts/experiments/stress_parity.py 30 2000
run blocks_bb13 python scripts/experiments/stress_observable_blocks.py 60 2000
QHBM_OBS_BLOCK_BITS=12 run blocks_bb12 python scripts/experiments/stress_observable_blocks.py 60 2000
