#!/bin/bash
# Adjoint tile of 2^12 against 2^13 amplitudes over the qubit count (developer tool; run via gpurun).
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); k=d["kernel_ms_per_step"]; print("%-40s step %9.2f  fwd %8.2f  adj %8.2f  obs %6.2f  passes %s/%s" % (sys.argv[1], d["ms_per_step"], k["forward"], k["adjoint"], k["apply_observable"], d["config"]["forward_passes"], d["config"]["adjoint_passes"]))'
for spec in "20 16 xxz 2048" "22 16 xxz 512" "24 16 xxz 128" "24 32 tfim 128" "26 16 xxz 32" "26 32 tfim 32" "28 16 xxz 16" "28 32 tfim 16"; do
  set -- $spec
  for k in 0 12 13; do
    python bench.py --qubits $1 --layers $2 --hamiltonian $3 --states-total $4 --steps 2 --warmup 1 --no-cpu-baseline --engine-option adjoint_tile_qubits=$k 2>&1 | python -c "$P" "n=$1 L=$2 $3 U=$4 adjK=$k"
  done
done
