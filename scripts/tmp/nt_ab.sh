#!/bin/bash
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for lib in "" "/root/repo/scripts/tmp/obsmod/lib_tilent.so"; do
  QHBM_ENGINE_LIB=$lib timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib'[-18:] or 'HEAD', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
done; done
