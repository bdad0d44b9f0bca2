#!/bin/bash
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
for o in "" "--engine-option tile_qubits=13" "--engine-option tile_qubits=13 --engine-option wide_last_pass=1"; do
for q in 20 21; do
S=$((4096 >> (q-20)))
python bench.py --qubits $q --layers 16 --states-total $S --steps 3 --warmup 1 --no-cpu-baseline $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xxz n=$q', d['config']['engine_options'], round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()}, d['config']['forward_passes'])"
done; done
