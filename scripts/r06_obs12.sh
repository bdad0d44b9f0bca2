#!/bin/bash
# Round 6: the block kernel in its two shapes (blocks of 2^13 under one workgroup of two halves per CU; blocks of 2^12 under
# two independent workgroups per CU), one box: parity tests of the block kernels under both, then config 4 and config 3 timed.
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_obs12
mkdir -p "$OUT"
cd "$R"
for bb in 12 13; do
  (time QHBM_OBS_BLOCK_BITS=$bb timeout 900 python -m pytest tests/test_observable_blocks_gpu.py tests/test_golden_large_gpu.py tests/test_bench_gpu.py -q -x --durations=5 -k "not ranks" ) > "$OUT/pytest_bb$bb.log" 2>&1
  tail -4 "$OUT/pytest_bb$bb.log"
done
for rep in 1 2; do
for bb in 13 12; do
  QHBM_OBS_BLOCK_BITS=$bb timeout 600 python bench.py --qubits 24 --layers 16 --states-total 32 --hamiltonian random512 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/c4_adj_bb${bb}_$rep.json" 2> "$OUT/c4_adj_bb${bb}_$rep.err"
  QHBM_OBS_BLOCK_BITS=$bb timeout 600 python bench.py --qubits 24 --layers 16 --states-total 32 --hamiltonian random512 --mode forward --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/c4_fwd_bb${bb}_$rep.json" 2> "$OUT/c4_fwd_bb${bb}_$rep.err"
  QHBM_OBS_BLOCK_BITS=$bb timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-mirror-step > "$OUT/c3_bb${bb}_$rep.json" 2> "$OUT/c3_bb${bb}_$rep.err"
  python - "$OUT" $bb $rep <<'PY'
import json,sys
out,bb,rep=sys.argv[1:]
for name in ("c4_adj","c4_fwd","c3"):
  f=f"{out}/{name}_bb{bb}_{rep}.json"
  try:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(name, "bb", bb, "rep", rep, round(d["ms_per_step"],2), "ms", {k: round(v,2) for k,v in d.get("kernel_ms_per_step",{}).items()})
  except Exception as e:
    print(f, "FAILED", e); print(open(f[:-5]+".err").read()[-800:])
PY
done
done
