#!/bin/bash
# Developer tool: bench.py on the config-3 circuit (512 states) under a list of engine options.
#   gpurun -- 'bash scripts/experiments/sweep_options.sh "adjoint_tile_qubits=13" "adjoint_full_diag_threshold=30" ...'
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); k=d["kernel_ms_per_step"]; print("%-44s step %7.2f  fwd %6.2f  adj %6.2f  obs %5.2f  passes %s/%s" % (",".join(d["config"]["engine_options"]) or "default", d["ms_per_step"], k["forward"], k["adjoint"], k["apply_observable"], d["config"]["forward_passes"], d["config"]["adjoint_passes"]))'
for o in "" "$@"; do
  args=""
  for kv in $o; do args="$args --engine-option $kv"; done
  python bench.py --states-total 512 --steps 5 --warmup 2 --no-cpu-baseline $args 2>&1 | python -c "$P"
done
