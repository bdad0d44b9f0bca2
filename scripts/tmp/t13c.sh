#!/bin/bash
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
for q in 22 23; do
python bench.py --qubits $q --layers 16 --states-total 64 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xxz n=$q', round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()}, d['config']['forward_passes'])"
done
python bench.py --qubits 24 --layers 16 --states-total 32 --hamiltonian random512 --steps 3 --warmup 1 --cpu-sample-states 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 adj', round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()}, d['parity_check']['ok'])"
python bench.py --qubits 24 --layers 16 --states-total 2 --hamiltonian random512 --mode shift --steps 1 --warmup 0 --cpu-sample-states 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 shift', round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()}, d['parity_check']['ok'])"
timeout 900 python -m pytest tests/test_golden_large_gpu.py -q -x -k "c4 or c5" 2>&1 | tail -2
