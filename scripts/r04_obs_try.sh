#!/bin/bash
# Round 4 A/B helper: config 4's adjoint step under a list of "ENV=... --engine-option ..." variants, one line each.
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_obs_try
mkdir -p "$OUT"
cd "$R"
C4="--qubits 24 --layers 16 --states-total 32 --hamiltonian random512 --steps 3 --warmup 1 --no-cpu-baseline"
i=0
while IFS= read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  envs=""; args=""
  for tok in $line; do case "$tok" in *=*) if [[ "$tok" == QHBM_* ]]; then envs="$envs $tok"; else args="$args --engine-option $tok"; fi;; esac; done
  env $envs timeout 600 python bench.py $C4 $args > "$OUT/v$i.json" 2> "$OUT/v$i.err"
  python - "$OUT/v$i.json" "$line" <<'PY'
import json,sys
try:
  d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("%-70s"%sys.argv[2], round(d["ms_per_step"],2), {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()})
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done
