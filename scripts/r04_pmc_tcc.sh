#!/bin/bash
# TCC hit / miss and fabric bytes of the observable kernel under env / option variants (one per stdin line)
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_tcc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
C4="${BENCH_ARGS:---qubits 24 --layers 16 --states-total 32 --hamiltonian random512} --steps 1 --warmup 1 --no-cpu-baseline"
i=0
while IFS= read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  args=""
  for tok in $line; do case "$tok" in QHBM_*=*) export "$tok";; *=*) args="$args --engine-option $tok";; esac; done
  for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    rm -rf "$OUT/p"; timeout 600 rocprofv3 --kernel-trace --pmc $C -d "$OUT/p" -o b --output-format csv -- python3 "$R/bench.py" $C4 $args > "$OUT/log" 2>&1
    python3 - "$OUT/p" "$line" <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(list); dur=[]
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
  for row in csv.DictReader(open(f)):
    if "observable" in row["Kernel_Name"] and "value_parts" not in row["Kernel_Name"]:
      agg[row["Counter_Name"]].append(float(row["Counter_Value"])); dur.append((int(row["End_Timestamp"])-int(row["Start_Timestamp"]))/1e6)
print("%-60s"%sys.argv[2], "ms %.1f"%(sum(dur)/max(1,len(dur))), {k:"%.4g"%(sum(v)/len(v)) for k,v in agg.items()})
PY
  done
  for tok in $line; do case "$tok" in QHBM_*=*) unset "${tok%%=*}";; esac; done
done
