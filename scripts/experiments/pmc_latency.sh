R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_lat; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS" "SQ_LEVEL_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o p --output-format csv -- python3 $R/scripts/experiments/one_step.py 20 16 256 xxz vqt > $OUT/p$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(out+"/p*/p_counter_collection.csv")):
  for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k,d in agg.items():
  if "pass_" not in k and "apply_obs" not in k: continue
  print(k)
  for c,v in sorted(d.items()): print(f"   {c:28s} {v:16.0f}  (dispatches {n[k][c]})")
PY
