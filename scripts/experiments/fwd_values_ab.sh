P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); k=d.get("kernel_ms_per_step") or {}; print("%-50s step %9.3f ms  fwd %8.2f obs %7.2f" % (sys.argv[1], d["ms_per_step"], k.get("forward",0), k.get("apply_observable",0)))'
for o in 0 1; do
python bench.py --no-cpu-baseline --qubits 20 --layers 16 --hamiltonian xxz --states-total 2048 --steps 3 --warmup 1 --mode forward --engine-option forward_values_from_observable=$o 2>&1 | python -c "$P" "C3 2048 fwd obsroute=$o"
python bench.py --no-cpu-baseline --qubits 24 --layers 16 --hamiltonian random512 --states-total 32 --steps 3 --warmup 1 --mode forward --engine-option forward_values_from_observable=$o 2>&1 | python -c "$P" "C4 32 fwd obsroute=$o"
python bench.py --no-cpu-baseline --qubits 28 --layers 32 --hamiltonian tfim --states-total 16 --steps 2 --warmup 1 --mode forward --engine-option forward_values_from_observable=$o 2>&1 | python -c "$P" "C5 16 fwd obsroute=$o"
python bench.py --no-cpu-baseline --qubits 16 --layers 8 --hamiltonian tfim --states-total 4096 --steps 5 --warmup 1 --mode forward --engine-option forward_values_from_observable=$o 2>&1 | python -c "$P" "n16 L8 tfim 4096 fwd obsroute=$o"
python bench.py --no-cpu-baseline --qubits 12 --layers 8 --hamiltonian tfim --states-total 1024 --steps 20 --warmup 3 --mode forward --engine-option forward_values_from_observable=$o 2>&1 | python -c "$P" "C2 fwd obsroute=$o"
done
