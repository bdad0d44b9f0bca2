#!/bin/bash
# Round 6: rocprofv3 kernel stats + counter passes of every BASELINE config shape and the QMHL step
# (gpurun from the repo root; then `python scripts/summarize_profile.py <tag>` per tag here).
#   bash scripts/r06_profiles.sh <git-head> [tag ...]      (no tags: all)
set -u
ulimit -c 0
HEAD=${1:?git head}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
declare -A CFG=(
  [r06_c2]="--qubits 12 --layers 8 --states-total 1024 --hamiltonian tfim"
  [r06_c3_4096]=""
  [r06_c3x3]="--hamiltonian xxz3"
  [r06_c4_adj]="--qubits 24 --layers 16 --states-total 32 --hamiltonian random512"
  [r06_c4_shift]="--qubits 24 --layers 16 --states-total 2 --hamiltonian random512 --mode shift --steps 1 --warmup 0"
  [r06_c5]="--qubits 28 --layers 32 --states-total 16 --hamiltonian tfim"
  [r06_qmhl]="--mode qmhl"
)
TAGS=("$@"); [ ${#TAGS[@]} -eq 0 ] && TAGS=(r06_c3_4096 r06_c2 r06_c3x3 r06_c4_adj r06_c4_shift r06_c5 r06_qmhl)
for T in "${TAGS[@]}"; do
  echo "== $T"; cd "$R"
  bash scripts/profile_bench.sh "$T" "$HEAD" ${CFG[$T]} 2>&1 | tail -2
done
