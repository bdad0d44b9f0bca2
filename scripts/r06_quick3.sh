#!/bin/bash
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_quick3
mkdir -p "$OUT"
cd "$R"
(time timeout 1500 python -m pytest tests/test_captured_gpu.py tests/test_bench_gpu.py tests/test_ebm_gpu.py tests/test_observable_blocks_gpu.py -q --durations=12 -x) > "$OUT/pytest.log" 2>&1
tail -22 "$OUT/pytest.log"
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
./scripts/experiments/micro/mfma_reduce.bin > "$OUT/mfma_reduce.txt" 2>&1; cat "$OUT/mfma_reduce.txt"
python scripts/experiments/profile_mirror_c2.py c2 > "$OUT/mirror_profile_c2.txt" 2>&1; head -60 "$OUT/mirror_profile_c2.txt" | cut -c1-180
