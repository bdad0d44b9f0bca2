R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/trace_qmhl; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT" -o t --output-format csv -- python3 "$R/scripts/experiments/qmhl_mirror_time.py" 2048 > "$OUT/log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/t_kernel_trace.csv", recursive=True):
  for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("void qhbm::", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", "")))
rows.sort()
# last step: find the last occurrence of the init pass
idx=[i for i,(t,n,d,g) in enumerate(rows) if n.startswith("pass_fwd_kernel")]
# print the last 40 qhbm kernels
sel=[r for r in rows if ("pass_" in r[1] or "apply_obs" in r[1] or "reduce" in r[1] or "parity" in r[1])]
for _, n, d, g in sel[-32:]: print(f"{n:44s} {d:9.3f} ms  grid {g}")
PY
