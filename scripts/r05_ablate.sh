#!/bin/bash
# Per-piece attribution of the adjoint sweep (round 5): one VQT step of config 3 per ablated library
# (scripts/experiments/ablate/build.py; their RESULTS are wrong, only time and instruction counts mean something),
# kernel time from --kernel-trace and SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU from a counter pass.
#   gpurun -- 'bash scripts/r05_ablate.sh <states> base no_reduce ...'
S=${1:-1024}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  OUT=$R/gpurun_out/r05_ablate/$v; rm -rf "$OUT"; mkdir -p "$OUT"
  export QHBM_ENGINE_LIB=$R/scripts/experiments/ablate/lib_$v.so
  [ "$v" = head ] && unset QHBM_ENGINE_LIB
  rocprofv3 --kernel-trace -d "$OUT/t" -o t --output-format csv -- python3 "$R/scripts/experiments/one_step.py" 20 16 $S xxz vqt > "$OUT/log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/c" -o t --output-format csv -- python3 "$R/scripts/experiments/one_step.py" 20 16 $S xxz vqt > "$OUT/logc" 2>&1
  python3 - "$OUT" "$v" <<'PY'
import csv, glob, sys
out, v = sys.argv[1], sys.argv[2]
def short(n): return n.replace("void qhbm::", "").replace("(anonymous namespace)::", "").split("(")[0]
rows = []
for f in glob.glob(out + "/t/**/t_kernel_trace.csv", recursive=True):
  for r in csv.DictReader(open(f)):
    n = short(r["Kernel_Name"])
    if "pass_" in n or "observable" in n: rows.append((int(r["Start_Timestamp"]), n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort(); rows = rows[len(rows) // 2:]
tot = {}
for _, n, d in rows: tot[n.split('<')[0]] = tot.get(n.split('<')[0], 0.0) + d
adj = [d for _, n, d in rows if n.startswith("pass_adjx")]
ctr = {}
for f in glob.glob(out + "/c/**/t_counter_collection.csv", recursive=True):
  for r in csv.DictReader(open(f)):
    n = short(r["Kernel_Name"])
    if n.startswith("pass_adjx"):
      ctr[r["Counter_Name"]] = ctr.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]) / 2  # two steps
print(f"{v:16s} fwd {tot.get('pass_fwd2_kernel',0)+tot.get('pass_fwd_kernel',0):7.2f} obs {tot.get('apply_observable_kernel',0):6.2f} adj {tot.get('pass_adjx_kernel',0):7.2f} ms | adj passes " + " ".join(f"{d:.2f}" for d in adj) +
      f" | adj VALU insts {ctr.get('SQ_INSTS_VALU',0)/1e9:.3f} G  SALU {ctr.get('SQ_INSTS_SALU',0)/1e9:.3f} G  VALU active/busy {4*ctr.get('SQ_ACTIVE_INST_VALU',0)/max(ctr.get('GRBM_GUI_ACTIVE',1)/8*1024,1):.3f}")
PY
done
