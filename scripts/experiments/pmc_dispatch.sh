#!/bin/bash
# Per-DISPATCH hardware counters of one VQT step on the C3 shard (developer tool; run via gpurun):
#   bash scripts/experiments/pmc_dispatch.sh "<counters>" <states> [engine option=value ...]
set -u
CTRS=${1:?counters}; STATES=${2:-512}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_dispatch; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS -d "$OUT/p" -o p --output-format csv -- python3 "$R/scripts/experiments/one_step.py" 20 16 $STATES xxz vqt "$@" > "$OUT/p.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.OrderedDict()
for f in sorted(glob.glob(out + "/p/**/p_counter_collection.csv", recursive=True)):
  for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void qhbm::", "")[:40]
    if "pass_" not in k and "apply_obs" not in k: continue
    d = rows.setdefault(int(r["Dispatch_Id"]), {"kernel": k, "grid": r.get("Grid_Size", r.get("Grid_Size_X", ""))})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(rows)
for i in ids[len(ids) // 2:]:
  d = rows[i]
  print(f"{d['kernel']:40s} grid {d['grid']:>10s} " + "  ".join(f"{c}={v:.4g}" for c, v in d.items() if c not in ("kernel", "grid")))
PY
