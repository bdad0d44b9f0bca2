#!/bin/bash
# Profiles one bench.py workload on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash scripts/profile_bench.sh <tag> <git-head> [bench.py args...]'
# e.g.  bash scripts/profile_bench.sh r02_c3_4096 $(git rev-parse --short HEAD)
#       bash scripts/profile_bench.sh r02_c2 abc1234 --qubits 12 --layers 8 --states-total 1024 --hamiltonian tfim
# Separate rocprofv3 runs: kernel stats; FETCH_SIZE; WRITE_SIZE (the two TCC counters do not fit one
# pass); three SQ passes (VALU / LDS / wait).  --pmc is never combined with another trace domain
# than --kernel-trace (this pool refuses that), and the program after `--` is python3 itself.
set -u
TAG=${1:?tag}; HEAD=${2:?git head}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-mirror-step $*"
echo "$HEAD" > "$OUT/git_head"
echo "python3 bench.py $ARGS" > "$OUT/command"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/bench_stats.log" 2>&1
grep "^{\"metric\"" "$OUT/bench_stats.log" | tail -1 > "$OUT/bench_line_profiled.json"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc$i" -o bench --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/pmc$i.log" 2>&1
done
# keep what scripts/summarize_profile.py reads; the raw traces are large
find "$OUT" -name "*_kernel_trace.csv" -size +8M -delete
du -sh "$OUT"; cat "$OUT/bench_line_profiled.json" | head -c 600
