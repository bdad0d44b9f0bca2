#!/bin/bash
# Round 6, last look: the bench tests and the default line with the mirror step, the randomized sweeps, the 8-GPU projection.
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_final
mkdir -p "$OUT"
cd "$R"
(time timeout 900 python -m pytest tests/test_bench_gpu.py tests/test_captured_gpu.py -q -x) > "$OUT/pytest.log" 2>&1
tail -6 "$OUT/pytest.log"
timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/c3_default.json" 2> "$OUT/c3_default.err"
python - "$OUT/c3_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("default", round(d["ms_per_step"],2), d["kernel_ms_per_step"], d["parity_check"]["ok"], d["cpu_baseline"]["value"], d["cpu_baseline"]["checker"], d.get("vqt_step_through_mirror"), d["roofline"].get("stored_profile_warning"))
PY
(time timeout 900 python scripts/experiments/stress_default_plans.py 12) > "$OUT/stress_default_plans.txt" 2>&1; tail -3 "$OUT/stress_default_plans.txt"
(time timeout 900 python scripts/experiments/stress_measure.py 12) > "$OUT/stress_measure.txt" 2>&1; tail -3 "$OUT/stress_measure.txt"
(time timeout 900 python scripts/experiments/stress_api_sizes.py 16 20) > "$OUT/stress_api_sizes.txt" 2>&1; tail -7 "$OUT/stress_api_sizes.txt"
(time timeout 600 python scripts/experiments/stress_observable_blocks.py 8) > "$OUT/stress_observable_blocks.txt" 2>&1; tail -3 "$OUT/stress_observable_blocks.txt"
timeout 600 python3 scripts/projection_8gpu.py > "$OUT/projection_8gpu.json" 2> "$OUT/projection_8gpu.err"; head -c 1200 "$OUT/projection_8gpu.json"
