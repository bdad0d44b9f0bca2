#!/bin/bash
# Adjoint plan chosen among the scheduler's candidate orders by the time model (adjoint_plan_search = 1) against
# the proxy's first choice (0) (developer tool; run via gpurun).
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); k=d["kernel_ms_per_step"]; print("%-44s step %9.2f  fwd %8.2f  adj %8.2f  obs %6.2f  passes %s/%s" % (sys.argv[1], d["ms_per_step"], k["forward"], k["adjoint"], k["apply_observable"], d["config"]["forward_passes"], d["config"]["adjoint_passes"]))'
for spec in "22 16 xxz 512" "24 32 tfim 128" "26 32 tfim 32" "28 32 tfim 16" "24 24 tfim 128" "20 32 xxz 1024"; do
  set -- $spec
  for k in 0 1; do
    python bench.py --qubits $1 --layers $2 --hamiltonian $3 --states-total $4 --steps 2 --warmup 1 --no-cpu-baseline --engine-option adjoint_plan_search=$k 2>&1 | python -c "$P" "n=$1 L=$2 $3 U=$4 search=$k"
  done
done
