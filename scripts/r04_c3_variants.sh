#!/bin/bash
# Round 4, VERDICT #5: config 3's step under engine-option variants -- time (un-profiled run) and SQ_INSTS_VALU per
# launch of the pass kernels (one rocprofv3 --pmc pass each).  One variant per stdin line ("name opt=val ..."):
#   gpurun -- "bash scripts/r04_c3_variants.sh < scripts/r04_c3_variants.txt"
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_c3_variants
mkdir -p "$OUT"
ARGS_EXTRA=${C3_ARGS:-}
while IFS= read -r line; do
  [ -z "$line" ] && continue
  name=${line%% *}; opts=""
  for tok in ${line#* }; do case "$tok" in *=*) opts="$opts --engine-option $tok";; esac; done
  cd "$R"
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $ARGS_EXTRA $opts > "$OUT/$name.json" 2> "$OUT/$name.err"
  cd /tmp && export TMPDIR=/tmp
  rm -rf "$OUT/p"; timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -d "$OUT/p" -o b --output-format csv -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline $ARGS_EXTRA $opts > "$OUT/pmc.log" 2>&1
  python3 - "$OUT/$name.json" "$OUT/p" "$line" <<'PY'
import json,sys,csv,glob,collections
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[2]+"/**/*counter_collection.csv", recursive=True):
  for row in csv.DictReader(open(f)):
    k=row["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","").replace("qhbm::","")
    if k.startswith("pass_") or "observable" in k: agg[k].append(float(row["Counter_Value"]))
valu={k:"%.3gG x%d"%(sum(v)/len(v)/1e9,len(v)//2) for k,v in agg.items()}
print("%-34s step %.1f ms  fwd %.1f obs %.1f adj %.1f  passes %s+%s  VALU insts per launch: %s"%(sys.argv[3][:34], d["ms_per_step"], d["kernel_ms_per_step"]["forward"], d["kernel_ms_per_step"]["apply_observable"], d["kernel_ms_per_step"]["adjoint"], d["config"]["forward_passes"], d["config"]["adjoint_passes"], valu))
PY
done
