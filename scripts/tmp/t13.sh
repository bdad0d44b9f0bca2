#!/bin/bash
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
for o in "" "--engine-option tile_qubits=13"; do
python bench.py --qubits 24 --layers 16 --states-total 2 --hamiltonian random512 --mode shift --steps 1 --warmup 0 --cpu-sample-states 1 $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shift', d['config']['engine_options'], round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()}, d['parity_check']['ok'])"
for q in 22 23; do
python bench.py --qubits $q --layers 16 --states-total 64 --steps 3 --warmup 1 --no-cpu-baseline $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xxz n=$q', d['config']['engine_options'], round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()})"
done; done
