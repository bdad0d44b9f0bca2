#!/bin/bash
# Forward tiles of 2^12 against 2^13 amplitudes on the XXZ chain at 20..23 qubits and on config 4's lean passes
# (developer tool, round 4: the rule "2^13 from 22 qubits on" in schedule.cpp build_plan comes from this).
#   gpurun -- 'bash scripts/experiments/fwd_tile_ab.sh'
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["config"]["engine_options"], round(d["ms_per_step"],1), {k:round(v,1) for k,v in d["kernel_ms_per_step"].items()}, d["config"]["forward_passes"])'
for o in "--engine-option tile_qubits=12" "--engine-option tile_qubits=13"; do
  for q in 20 21 22 23; do
    S=$((4096 >> (2 * (q - 20)))); [ $S -lt 64 ] && S=64
    python bench.py --qubits $q --layers 16 --states-total $S --steps 3 --warmup 1 --no-cpu-baseline $o 2>/dev/null | python -c "$show" "xxz n=$q states=$S"
  done
  python bench.py --qubits 24 --layers 16 --states-total 2 --hamiltonian random512 --mode shift --steps 1 --warmup 0 --no-cpu-baseline $o 2>/dev/null | python -c "$show" "config 4, parameter shift"
done
