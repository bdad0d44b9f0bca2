#!/bin/bash
# Every BASELINE config shape on one GPU at HEAD (developer tool; run via gpurun): one line per run into
# gpurun_out/configs_r03.txt -- step, kernel times, pass counts.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/configs_r03.txt
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); k=d.get("kernel_ms_per_step") or {}; c=d["config"]; print("%-58s step %9.3f ms  value %.4g %s  fwd %8.2f  obs %7.2f  adj %8.2f  passes %s/%s  parity %s" % (sys.argv[1], d["ms_per_step"], d["value"], d["unit"], k.get("forward", 0), k.get("apply_observable", 0), k.get("adjoint", 0), c.get("forward_passes"), c.get("adjoint_passes"), (d.get("parity_check") or {}).get("ok")))'
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" 2>&1 | python -c "$P" "$name" >> "$OUT" || echo "$name FAILED" >> "$OUT"; }
: > "$OUT"
run "C2 n=12 L=8 tfim 1024 states vqt"            --qubits 12 --layers 8 --hamiltonian tfim --states-total 1024 --steps 20 --warmup 3
run "C3 n=20 L=16 xxz 512 states vqt"             --qubits 20 --layers 16 --hamiltonian xxz --states-total 512 --steps 5 --warmup 2
run "C3 n=20 L=16 xxz 4096 states vqt"            --qubits 20 --layers 16 --hamiltonian xxz --states-total 4096 --steps 3 --warmup 1
run "C3 n=20 L=16 xxz 4096 states forward"        --qubits 20 --layers 16 --hamiltonian xxz --states-total 4096 --steps 3 --warmup 1 --mode forward
run "C4 n=24 L=16 random512 32 states adjoint"    --qubits 24 --layers 16 --hamiltonian random512 --states-total 32 --steps 3 --warmup 1
run "C4 same, adjoint_relabel=0"                  --qubits 24 --layers 16 --hamiltonian random512 --states-total 32 --steps 3 --warmup 1 --engine-option adjoint_relabel=0
run "C5 n=28 L=32 tfim 16 states vqt"             --qubits 28 --layers 32 --hamiltonian tfim --states-total 16 --steps 2 --warmup 1
run "C5 same, adjoint_relabel=0"                  --qubits 28 --layers 32 --hamiltonian tfim --states-total 16 --steps 2 --warmup 1 --engine-option adjoint_relabel=0
run "C5 n=28 L=32 tfim 32 states (one rank's share) vqt" --qubits 28 --layers 32 --hamiltonian tfim --states-total 32 --steps 2 --warmup 1
cat "$OUT"
