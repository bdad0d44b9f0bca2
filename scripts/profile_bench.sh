#!/bin/bash
# Profiles the default bench workload on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash scripts/profile_bench.sh'
# Three separate rocprofv3 runs (kernel stats; FETCH_SIZE; WRITE_SIZE -- the two TCC counters do
# not fit one pass, and --pmc must not be combined with other trace domains on this pool).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/bench_stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o bench --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o bench --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/bench_write.log" 2>&1
grep "^{\"metric\"" "$OUT/bench_stats.log" | tail -1 > "$OUT/bench_line_profiled.json"
ls -R "$OUT" | head -40
