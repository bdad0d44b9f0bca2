#!/bin/bash
# A/B of hardware counters on the C3 shard (developer tool; run via gpurun):
#   bash scripts/experiments/pmc_ab.sh <tag> "<counters>" <states> [engine option=value ...]
# one rocprofv3 --pmc pass per counter of scripts/experiments/one_step.py (a VQT step); prints per-kernel counter sums per dispatch.
set -u
TAG=${1:?tag}; CTRS=${2:?counters}; STATES=${3:-512}; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_ab_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# one counter per pass (FETCH_SIZE with WRITE_SIZE in one pass "exceeds the capabilities of the hardware": the
# profiler aborts and then hangs in its finaliser -- hence also the timeout)
for C in $CTRS; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/p/$C" -o p --output-format csv -- python3 "$R/scripts/experiments/one_step.py" 20 16 $STATES xxz vqt "$@" > "$OUT/p_$C.log" 2>&1 || echo "pass $C failed (see $OUT/p_$C.log)"
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in sorted(glob.glob(out + "/p/**/p_counter_collection.csv", recursive=True)):
  for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void qhbm::", "")[:48]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k, d in agg.items():
  if "pass_" not in k and "apply_obs" not in k: continue
  print(f"{tag:28s} {k:48s} dispatches {len(disp[k]):3d} " + "  ".join(f"{c}={v / len(disp[k]):.4g}" for c, v in sorted(d.items())))
PY
