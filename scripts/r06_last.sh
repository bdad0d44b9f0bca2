#!/bin/bash
# Round 6, last call: smoke() and the default bench line as the driver runs them, on the final tree.
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_last
mkdir -p "$OUT"
cd "$R"
(time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')") > "$OUT/smoke.log" 2>&1; tail -3 "$OUT/smoke.log"
(time python bench.py --steps 20 --warmup 5) > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
tail -1 "$OUT/bench_default.json" | cut -c1-400
tail -3 "$OUT/bench_default.err"
