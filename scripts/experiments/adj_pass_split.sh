#!/bin/bash
# Per-launch durations of config 3's adjoint passes with the compute or the tile I/O compiled out (developer tool,
# round 4): is a pass the SUM of its memory phase and its arithmetic, or their maximum?
#   python scripts/experiments/ablate/build.py base no_instances adj_no_io   (here, on the CPU box)
#   gpurun -- 'bash scripts/experiments/adj_pass_split.sh [states] [ENV=value ...]'
S=${1:-1024}; shift
for kv in "$@"; do export "$kv"; done
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-base no_instances adj_no_io}; do
  OUT=$R/gpurun_out/adj_pass_split/$v; rm -rf "$OUT"; mkdir -p "$OUT"
  export QHBM_ENGINE_LIB=$R/scripts/experiments/ablate/lib_$v.so
  [ "$v" = head ] && unset QHBM_ENGINE_LIB
  rocprofv3 --kernel-trace -d "$OUT" -o t --output-format csv -- python3 "$R/scripts/experiments/one_step.py" 20 16 $S xxz vqt > "$OUT/log" 2>&1
  echo "== $v ($S states)"
  python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/t_kernel_trace.csv", recursive=True):
  for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("void qhbm::", "").replace("(anonymous namespace)::", "").split("(")[0]
    if "pass_" in n or "observable" in n: rows.append((int(r["Start_Timestamp"]), n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", "")))
rows.sort()
half = len(rows) // 2
tot = {}
for _, n, d, g in rows[half:]:
  print(f"  {n:36s} {d:9.3f} ms  grid {g}")
  tot[n.split('<')[0]] = tot.get(n.split('<')[0], 0.0) + d
print("  totals:", {k: round(v, 2) for k, v in tot.items()})
PY
done
