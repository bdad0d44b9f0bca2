#!/bin/bash
# Round 6: the block kernel with its partner blocks staged dword by dword (buffer_load_dword x 4 + ds_write_addtid_b32 x 4 per
# 16 bytes, -DQHBM_OBS_ADDTID=1) against the shipped staging (buffer_load_dwordx4 + ds_write_b128), one box.
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_addtid
mkdir -p "$OUT"
cd "$R"
LIB=${1:-scripts/tmp/lib_addtid.so}
(time QHBM_ENGINE_LIB=$R/$LIB timeout 900 python -m pytest tests/test_observable_blocks_gpu.py tests/test_golden_large_gpu.py -q -x --durations=5) > "$OUT/pytest_addtid.log" 2>&1
tail -4 "$OUT/pytest_addtid.log"
bash scripts/r05_ab.sh addtid_c4adj 2 "--qubits 24 --layers 16 --states-total 32 --hamiltonian random512 --steps 3 --warmup 1" head $LIB
bash scripts/r05_ab.sh addtid_c4fwd 2 "--qubits 24 --layers 16 --states-total 32 --hamiltonian random512 --mode forward --steps 3 --warmup 1" head $LIB
bash scripts/r05_ab.sh addtid_c3x3 1 "--hamiltonian xxz3 --steps 3 --warmup 1 --no-mirror-step" head $LIB
