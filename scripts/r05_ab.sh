#!/bin/bash
# A/B of engine libraries on ONE box (box-to-box spread is 4 %): bench.py per library, alternating, N rounds.
#   gpurun -- 'bash scripts/r05_ab.sh <tag> <rounds> "<bench args>" libA.so libB.so ...'   ("head" = the in-tree library)
TAG=${1:?tag}; ROUNDS=${2:?rounds}; ARGS=${3:?bench args}; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ab_$TAG; mkdir -p "$OUT"
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    name=$(basename "$lib" .so)
    if [ "$lib" = head ]; then unset QHBM_ENGINE_LIB; else export QHBM_ENGINE_LIB=$R/$lib; fi
    python3 "$R/bench.py" --no-cpu-baseline $ARGS 2>"$OUT/$name.$r.err" | grep '^{"metric"' | tail -1 > "$OUT/$name.$r.json"
    python3 - "$OUT/$name.$r.json" "$name" <<'PY'
import json, sys
try:
  d = json.load(open(sys.argv[1]))
  k = d.get("kernel_ms_per_step", {})
  print(f"{sys.argv[2]:24s} step {d['ms_per_step']:8.2f} ms  fwd {k.get('forward', 0):7.2f}  obs {k.get('apply_observable', 0):6.2f}  adj {k.get('adjoint', 0):7.2f}  parity {d.get('parity_check', {}).get('ok')}")
except Exception as e:
  print(sys.argv[2], "FAILED", e)
PY
  done
done
