P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); k=d["kernel_ms_per_step"]; print("%-40s step %9.2f  fwd %8.2f  adj %8.2f  obs %6.2f  passes %s/%s" % (sys.argv[1], d["ms_per_step"], k["forward"], k["adjoint"], k["apply_observable"], d["config"].get("forward_passes"), d["config"].get("adjoint_passes")))'
for k in 0 12 13; do
  python bench.py --mode qmhl --steps 2 --warmup 1 --no-cpu-baseline --engine-option adjoint_tile_qubits=$k 2>&1 | python -c "$P" "qmhl adjK=$k"
done
for k in 0 12 13; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --engine-option adjoint_tile_qubits=$k 2>&1 | python -c "$P" "c3 adjK=$k"
done
for k in 0 13; do
  python bench.py --qubits 28 --layers 32 --states-total 16 --hamiltonian tfim --steps 2 --warmup 1 --no-cpu-baseline --engine-option adjoint_tile_qubits=$k 2>&1 | python -c "$P" "c5 adjK=$k"
done
