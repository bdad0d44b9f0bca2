#!/bin/bash
# The two-level lambda = O psi sweep at config 5's shape (28 qubits, TFIM): launches of apply_observable_kernel with the far
# windows off / on, from a kernel trace (developer tool; run via gpurun).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for o in 0 -1; do
  OUT=$R/gpurun_out/r05_far/$o; rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --kernel-trace -d "$OUT" -o t --output-format csv -- python3 "$R/bench.py" --qubits 28 --layers 32 --states-total 16 --hamiltonian tfim --steps 1 --warmup 1 --no-cpu-baseline --engine-option observable_far_windows=$o > "$OUT/log" 2>&1
  python3 - "$OUT" "$o" <<'PY'
import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/t_kernel_trace.csv", recursive=True):
  for r in csv.DictReader(open(f)):
    if "apply_observable" in r["Kernel_Name"]: rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"].split("apply_observable_kernel")[1][:22], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
rows.sort(); n=len(rows)//2; rows=rows[-n:]
print("far_windows =", sys.argv[2], " launches of the last step:", " ".join(f"{n}:{d:.1f}ms" for _,n,d in rows), " state-sized transfers at 8 TB/s per ms: 34.4 GB = 4.3 ms")
PY
done
