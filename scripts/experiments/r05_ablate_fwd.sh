#!/bin/bash
# Per-piece attribution of the FORWARD sweep (round 5): one forward call of config 3 per ablated library
# (scripts/experiments/ablate/build.py fwd2_*; wrong results, only time and counters mean something).
#   gpurun -- 'bash scripts/experiments/r05_ablate_fwd.sh <states> head fwd2_no_instances ...'
S=${1:-1024}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  OUT=$R/gpurun_out/r05_ablate_fwd/$v; rm -rf "$OUT"; mkdir -p "$OUT"
  export QHBM_ENGINE_LIB=$R/scripts/experiments/ablate/lib_$v.so
  [ "$v" = head ] && unset QHBM_ENGINE_LIB
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/c" -o t --output-format csv -- python3 "$R/scripts/experiments/one_step.py" 20 16 $S xxz fwd > "$OUT/logc" 2>&1
  python3 - "$OUT" "$v" <<'PY'
import csv, glob, sys
out, v = sys.argv[1], sys.argv[2]
def short(n): return n.replace("void qhbm::", "").replace("(anonymous namespace)::", "").split("(")[0]
rows = []
for f in glob.glob(out + "/c/**/t_kernel_trace.csv", recursive=True):
  for r in csv.DictReader(open(f)):
    n = short(r["Kernel_Name"])
    if "pass_fwd" in n: rows.append((int(r["Start_Timestamp"]), n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Dispatch_Id"]))
rows.sort(); rows = rows[len(rows) // 2:]
ids = {r[3] for r in rows}
ctr = {}
for f in glob.glob(out + "/c/**/t_counter_collection.csv", recursive=True):
  for r in csv.DictReader(open(f)):
    if r["Dispatch_Id"] in ids: ctr[r["Counter_Name"]] = ctr.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
cyc = ctr.get("GRBM_GUI_ACTIVE", 0) / 8
print(f"{v:20s} fwd {sum(d for _, _, d, _ in rows):7.2f} ms | passes " + " ".join(f"{n.split('_kernel')[0][5:]}{n[n.index('<'):]}:{d:.2f}" for _, n, d, _ in rows) +
      f" | {cyc/1e6:.1f} M cycles  VALU insts {ctr.get('SQ_INSTS_VALU',0)/1e9:.3f} G  SALU {ctr.get('SQ_INSTS_SALU',0)/1e9:.3f} G  VALU active {4*ctr.get('SQ_ACTIVE_INST_VALU',0)/max(cyc*1024,1):.3f}")
PY
done
