#!/bin/bash
# Round 6: the partner-row reads of the block kernel through absolute LDS addresses (no v_add_u32 of the array's link-time 0
# per row) against the previous build, one box.
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_abs
mkdir -p "$OUT"
cd "$R"
PREV=${1:-scripts/tmp/lib_prev.so}
(time timeout 900 python -m pytest tests/test_observable_blocks_gpu.py tests/test_golden_large_gpu.py -q -x) > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log"
bash scripts/r05_ab.sh abs_c4adj 2 "--qubits 24 --layers 16 --states-total 32 --hamiltonian random512 --steps 3 --warmup 1" $PREV head
bash scripts/r05_ab.sh abs_c4fwd 2 "--qubits 24 --layers 16 --states-total 32 --hamiltonian random512 --mode forward --steps 3 --warmup 1" $PREV head
bash scripts/r05_ab.sh abs_c3x3 2 "--hamiltonian xxz3 --steps 3 --warmup 1 --no-mirror-step" $PREV head
bash scripts/r05_ab.sh abs_c4shift 1 "--qubits 24 --layers 16 --states-total 2 --hamiltonian random512 --mode shift --steps 1 --warmup 0" $PREV head
