#!/bin/bash
# Round 4: counters of the observable kernels on config 4's shape (gpurun from the repo root):
#   bash scripts/r04_pmc_obs.sh <tag> [bench.py args / --engine-option ...]
set -u
ulimit -c 0
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--qubits 24 --layers 16 --states-total 32 --hamiltonian random512 --steps 2 --warmup 1 --no-cpu-baseline $*"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc$i" -o bench --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/pmc$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
  seen = set()
  for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0][-60:]
    if "observable" not in k:
      continue
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (f, row.get("Dispatch_Id"))
    if key not in seen:
      seen.add(key); calls[(f, k)] += 1
for k, d in agg.items():
  n = max(c for (f, kk), c in calls.items() if kk == k)
  print(k, "dispatches per pass:", n)
  for c, v in sorted(d.items()):
    print("   %-24s %.4g per dispatch" % (c, v / n))
PY
du -sh "$OUT"; find "$OUT" -name "*.csv" -size +4M -delete
