#!/bin/bash
# Round 4: the block-grouped observable kernel against the gather kernel (gpurun from the repo root).
set -u
ulimit -c 0   # (a GPU fault must not fill the box with core dumps)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_obs_ab
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_observable_blocks_gpu.py -x -q > "$OUT/pytest_blocks.log" 2>&1
tail -5 "$OUT/pytest_blocks.log"
C4="--qubits 24 --layers 16 --states-total 32 --hamiltonian random512 --steps 3 --warmup 1 --no-cpu-baseline"
for K in 0 1; do for X in 0 1; do
  timeout 600 python bench.py $C4 --engine-option observable_kernel=$K --engine-option observable_xcd_states=$X > "$OUT/c4_adj_k${K}_x${X}.json" 2> "$OUT/c4_adj_k${K}_x${X}.err"
  python - "$OUT/c4_adj_k${K}_x${X}.json" <<'PY'
import json,sys
try:
  d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(d["ms_per_step"],2), {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done; done
C3="--steps 5 --warmup 2 --no-cpu-baseline"
for K in 0 1; do
  timeout 600 python bench.py $C3 --engine-option observable_kernel=$K > "$OUT/c3_k${K}.json" 2> "$OUT/c3_k${K}.err"
  python - "$OUT/c3_k${K}.json" <<'PY'
import json,sys
try:
  d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(d["ms_per_step"],2), {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
for K in 0 1; do
  timeout 900 python bench.py --qubits 24 --layers 16 --states-total 2 --hamiltonian random512 --mode shift --steps 1 --warmup 0 --no-cpu-baseline --engine-option observable_kernel=$K > "$OUT/c4_shift_k${K}.json" 2> "$OUT/c4_shift_k${K}.err"
  python - "$OUT/c4_shift_k${K}.json" <<'PY'
import json,sys
try:
  d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(d["ms_per_step"],2), {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
