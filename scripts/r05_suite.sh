#!/bin/bash
# Round 5: the whole GPU suite, then the bench lines of the BASELINE shapes (gpurun from the repo root).
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_suite
mkdir -p "$OUT"
cd "$R"
if [ -z "${SKIP_TESTS:-}" ]; then  # (SKIP_TESTS=1: the bench lines alone, e.g. on a second box)
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
tail -5 "$OUT/pytest_gpu.log"
fi
run() {  # name, args...
  local name=$1; shift
  timeout 900 python bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
  python - "$OUT/$name.json" <<'PY'
import json,sys
try:
  d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("%-14s"%sys.argv[1].split('/')[-1][:-5], round(d["ms_per_step"],2), "ms", "%.3g evals/s"%d["value"], {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()}, d.get("parity_check",{}).get("ok"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
run c3 --steps 5 --warmup 2
run c3_fwd --steps 5 --warmup 2 --mode forward
run c3x3 --steps 5 --warmup 2 --hamiltonian xxz3
run c3x3_fwd --steps 5 --warmup 2 --hamiltonian xxz3 --mode forward
run c2 --qubits 12 --layers 8 --states-total 1024 --hamiltonian tfim --steps 20 --warmup 5
run c4_adj --qubits 24 --layers 16 --states-total 32 --hamiltonian random512 --steps 3 --warmup 1 --cpu-sample-states 4
run c4_shift --qubits 24 --layers 16 --states-total 2 --hamiltonian random512 --mode shift --steps 1 --warmup 0 --cpu-sample-states 1
# (28 qubits: the C oracle needs ~an hour per state at depth 32; parity at this size is tests/test_golden_large_gpu.py's c5 fixtures)
run c5 --qubits 28 --layers 32 --states-total 16 --hamiltonian tfim --steps 2 --warmup 1 --no-cpu-baseline
run qmhl --mode qmhl --steps 3 --warmup 1 --cpu-sample-states 8
