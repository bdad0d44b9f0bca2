#!/bin/bash
# Round 6, first look: the new GPU tests, the mirror lines, config 4 as stated with and without the shared prefix.
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_quick
mkdir -p "$OUT"
cd "$R"
(time timeout 1500 python -m pytest tests/test_captured_gpu.py tests/test_engine_gpu.py tests/test_plan_switch_sizes_gpu.py tests/test_bench_gpu.py -q --durations=25 -k "captured or replayed or capture or prefix or plan_switch or 21_to_23 or 25_to_27 or mirror or tfq_compat or entry_points or bench") > "$OUT/pytest_new.log" 2>&1
tail -45 "$OUT/pytest_new.log"
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
for sh in 1 0; do
  timeout 900 python bench.py --qubits 24 --layers 16 --states-total 2 --hamiltonian random512 --mode shift --steps 1 --warmup 0 --no-cpu-baseline --engine-option shift_prefix_sharing=$sh > "$OUT/c4_shift_share$sh.json" 2> "$OUT/c4_shift_share$sh.err"
  python - "$OUT/c4_shift_share$sh.json" <<'PY'
import json,sys
try:
  d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(d["ms_per_step"],1), "ms", {k:round(v,1) for k,v in d["kernel_ms_per_step"].items()})
except Exception as e: print(sys.argv[1], "FAILED", e); print(open(sys.argv[1][:-5]+".err").read()[-1500:])
PY
done
