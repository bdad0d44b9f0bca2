#!/bin/bash
# Dynamic instruction mix of one bench step (VERDICT r4 #1b): the SQ_INSTS_VALU_* class counters next to
# SQ_INSTS_VALU, per launch of every kernel.   gpurun -- 'bash scripts/r05_instmix.sh <tag> [bench args]'
set -u
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/instmix_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline $*"
rocprofv3 -L > "$OUT/counter_list.txt" 2>&1
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
         "SQ_INSTS_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc$i" -o bench --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/pmc$i.log" 2>&1
  tail -3 "$OUT/pmc$i.log" | cut -c1-300
done
python3 - "$OUT" <<'PY'
import sys, os, pandas as pd
out = sys.argv[1]
rows = {}
for i in range(1, 9):
  p = os.path.join(out, f"pmc{i}", "bench_counter_collection.csv")
  if not os.path.exists(p): continue
  d = pd.read_csv(p)
  d["Name"] = d.Kernel_Name.str.replace("(anonymous namespace)::", "", regex=False).str.split("(").str[0].str.replace("void ", "").str.replace("qhbm::", "")
  g = d.groupby(["Name", "Counter_Name"]).agg(calls=("Dispatch_Id", "nunique"), total=("Counter_Value", "sum"))
  for (n, c), r in g.iterrows():
    rows.setdefault(n, {})[c] = r.total / r.calls
    rows[n]["calls"] = int(r.calls)
df = pd.DataFrame(rows).T
df.to_csv(os.path.join(out, "instmix.csv"))
pd.set_option("display.width", 250); pd.set_option("display.max_columns", 50)
print(df.to_string())
PY
find "$OUT" -name "*_kernel_trace.csv" -size +4M -delete
find "$OUT" -name "bench_counter_collection.csv" -size +16M -delete
du -sh "$OUT"
