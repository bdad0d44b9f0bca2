#!/bin/bash
# Hardware-counter passes of one forward or VQT step on the C3 shard (developer tool; run via gpurun):
#   bash scripts/experiments/pmc.sh fwd|vqt [states] ; results under gpurun_out/pmc_<mode>/
set -u
MODE=${1:-fwd}; STATES=${2:-128}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$MODE
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d "$OUT/p$i" -o p --output-format csv -- python3 "$R/scripts/experiments/one_step.py" 20 16 $STATES xxz $MODE > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in sorted(glob.glob(out + "/p*/p_counter_collection.csv")):
  seen = set()
  for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (f, r["Dispatch_Id"]) not in seen:
      seen.add((f, r["Dispatch_Id"]))
  for k in {r["Kernel_Name"].split("(")[0][:60] for r in csv.DictReader(open(f))}:
    calls[k] = len({r["Dispatch_Id"] for r in csv.DictReader(open(f)) if r["Kernel_Name"].split("(")[0][:60] == k})
for k, d in agg.items():
  if "pass_" not in k and "apply_obs" not in k: continue
  print(k, "dispatches", calls[k])
  for c, v in sorted(d.items()): print(f"   {c:32s} {v:16.0f}")
PY
