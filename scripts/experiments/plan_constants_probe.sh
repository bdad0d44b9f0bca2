#!/bin/bash
# Developer probe: adjoint time of config 3 (1024 states) and config 4 (32 states) for different constants of
# the pass-order cost model (QHBM_PLAN_KFIXED / QHBM_PLAN_KMEMORY, read by schedule.cpp at plan time):
#   bash scripts/experiments/plan_constants_probe.sh "6 17" "6 10" ...
for cfg in "$@"; do set -- $cfg; echo -n "fixed=$1 mem=$2: "
  QHBM_PLAN_KFIXED=$1 QHBM_PLAN_KMEMORY=$2 python scripts/experiments/vqt_time.py 1024 2>&1 | grep -v amdgpu | tr '\n' ' '
  QHBM_PLAN_KFIXED=$1 QHBM_PLAN_KMEMORY=$2 python bench.py --qubits 24 --layers 16 --hamiltonian random512 --states-total 32 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 adj', round(l['kernel_ms_per_step']['adjoint'],1), l['config']['adjoint_passes'])"
done
