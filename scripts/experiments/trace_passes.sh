#!/bin/bash
# Per-launch durations of one VQT step on the C3 shard under rocprofv3 --kernel-trace (developer tool):
#   bash scripts/experiments/trace_passes.sh [states] [engine option=value ...]
S=${1:-1024}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/trace_passes; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT" -o t --output-format csv -- python3 "$R/scripts/experiments/one_step.py" 20 16 $S xxz vqt "$@" > "$OUT/log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/t_kernel_trace.csv", recursive=True):
  for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("void qhbm::", "").split("(")[0]
    if "pass_" in n or "apply_obs" in n: rows.append((int(r["Start_Timestamp"]), n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", "")))
rows.sort()
half = len(rows) // 2
for _, n, d, g in rows[half:]: print(f"{n:36s} {d:9.3f} ms  grid {g}")
PY
