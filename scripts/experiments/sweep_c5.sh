P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); k=d["kernel_ms_per_step"]; print("%-44s step %7.2f  fwd %6.2f  adj %6.2f  obs %5.2f  passes %s/%s" % (",".join(d["config"]["engine_options"]) or "default", d["ms_per_step"], k["forward"], k["adjoint"], k["apply_observable"], d["config"]["forward_passes"], d["config"]["adjoint_passes"]))'
for o in "" "tile_qubits=12" "tile_qubits=14" "adjoint_tile_qubits=13"; do
  args=""; for kv in $o; do args="$args --engine-option $kv"; done
  python bench.py --qubits 28 --layers 32 --hamiltonian tfim --states-total 16 --steps 2 --warmup 1 --no-cpu-baseline $args 2>&1 | python -c "$P"
done
