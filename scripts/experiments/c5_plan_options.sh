#!/bin/bash
# Config 5's shape (28 qubits, depth 32, TFIM, 16 states) under plan options (developer tool, round 4):
# default 1473 ms; adjoint_tile_qubits=13 1461; tile_qubits=14 1570; tile_qubits=12 1493; adjoint_plan_search=0 1466.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
for o in "" "--engine-option adjoint_tile_qubits=13" "--engine-option tile_qubits=14" "--engine-option tile_qubits=12" "--engine-option adjoint_plan_search=0"; do
  timeout 600 python bench.py --qubits 28 --layers 32 --states-total 16 --hamiltonian tfim --steps 2 --warmup 1 --no-cpu-baseline $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', d['config']['engine_options'], round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()}, d['config']['forward_passes'], d['config']['adjoint_passes'])"
done
