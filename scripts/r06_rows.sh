#!/bin/bash
# Round 6: the block kernel with the halves of its workgroup splitting the block's ROWS (every mask on both halves, four pairs
# per thread; option observable_split_rows) against the shipped split by masks, one box.
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_rows
mkdir -p "$OUT"
cd "$R"
(time QHBM_OBS_SPLIT_ROWS=1 timeout 900 python -m pytest tests/test_observable_blocks_gpu.py tests/test_golden_large_gpu.py -q -x --durations=5) > "$OUT/pytest_rows.log" 2>&1
tail -4 "$OUT/pytest_rows.log"
for rep in 1 2; do
for sr in 0 1; do
  QHBM_OBS_SPLIT_ROWS=$sr timeout 600 python bench.py --qubits 24 --layers 16 --states-total 32 --hamiltonian random512 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/c4_adj_sr${sr}_$rep.json" 2> "$OUT/c4_adj_sr${sr}_$rep.err"
  QHBM_OBS_SPLIT_ROWS=$sr timeout 600 python bench.py --qubits 24 --layers 16 --states-total 32 --hamiltonian random512 --mode forward --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/c4_fwd_sr${sr}_$rep.json" 2> "$OUT/c4_fwd_sr${sr}_$rep.err"
  QHBM_OBS_SPLIT_ROWS=$sr timeout 600 python bench.py --hamiltonian xxz3 --mode forward --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/c3x3_fwd_sr${sr}_$rep.json" 2> "$OUT/c3x3_fwd_sr${sr}_$rep.err"
  python - "$OUT" $sr $rep <<'PY'
import json,sys
out,sr,rep=sys.argv[1:]
for name in ("c4_adj","c4_fwd","c3x3_fwd"):
  f=f"{out}/{name}_sr{sr}_{rep}.json"
  try:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(name, "split_rows", sr, "rep", rep, round(d["ms_per_step"],2), "ms", {k: round(v,2) for k,v in d.get("kernel_ms_per_step",{}).items()})
  except Exception as e:
    print(f, "FAILED", e); print(open(f[:-5]+".err").read()[-800:])
PY
done
done
