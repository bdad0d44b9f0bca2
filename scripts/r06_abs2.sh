#!/bin/bash
# Round 6: the LDS exchanges of the paired forward kernel and the exchange adjoint kernel through absolute LDS addresses
# (round_load0 / round_store0: no v_add_u32 of the array's link-time 0 per access) against the previous build, one box.
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_abs2
mkdir -p "$OUT"
cd "$R"
PREV=${1:-scripts/tmp/lib_prev.so}
(time timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_golden_gpu.py tests/test_golden_large_gpu.py tests/test_default_plans_gpu.py -q -x) > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log"
bash scripts/r05_ab.sh abs2_c3 3 "--steps 5 --warmup 2 --no-mirror-step" $PREV head
bash scripts/r05_ab.sh abs2_c5 1 "--qubits 28 --layers 32 --states-total 16 --hamiltonian tfim --steps 2 --warmup 1" $PREV head
bash scripts/r05_ab.sh abs2_qmhl 1 "--mode qmhl --steps 3 --warmup 1" $PREV head
bash scripts/r05_ab.sh abs2_c2 2 "--qubits 12 --layers 8 --states-total 1024 --hamiltonian tfim --steps 20 --warmup 5" $PREV head
