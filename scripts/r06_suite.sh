#!/bin/bash
# Round 6: the whole GPU suite, then the bench lines of the BASELINE shapes (gpurun from the repo root).
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_suite
mkdir -p "$OUT"
cd "$R"
if [ -z "${SKIP_TESTS:-}" ]; then  # (SKIP_TESTS=1: the bench lines alone, e.g. on a second box)
(time timeout 2400 python -m pytest tests -m gpu -x -q --durations=40) > "$OUT/pytest_gpu.log" 2>&1
tail -60 "$OUT/pytest_gpu.log"
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
for c in c1 c2 c3; do
  timeout 600 python bench.py --through-mirror $c --steps 20 --warmup 5 > "$OUT/mirror_$c.json" 2> "$OUT/mirror_$c.err"
  python - "$OUT/mirror_$c.json" <<'PY'
import json,sys
try:
  d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print(sys.argv[1].split('/')[-1], {k: (round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k.endswith("_ms") or k.endswith("_step") or "over_engine" in k or "bitwise" in k or "diff" in k})
except Exception as e: print(sys.argv[1], "FAILED", e); print(open(sys.argv[1][:-5]+".err").read()[-1500:])
PY
done
run c4_shift_noshare --qubits 24 --layers 16 --states-total 2 --hamiltonian random512 --mode shift --steps 1 --warmup 0 --no-cpu-baseline --engine-option shift_prefix_sharing=0
