#!/bin/bash
# Forward tile of 2^12 / 2^13 amplitudes, pairs on / off: forward-only calls of config 3 (developer tool; run via gpurun).
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); k=d["kernel_ms_per_step"]; print("%-44s step %9.2f  fwd %8.2f  obs %6.2f  passes %s" % (sys.argv[1], d["ms_per_step"], k["forward"], k["apply_observable"], d["config"].get("forward_passes")))'
for o in "tile_qubits=0" "tile_qubits=12" "tile_qubits=13" "tile_qubits=12 wide_last_pass=0" "forward_pairs=0"; do
  E=""; for kv in $o; do E="$E --engine-option $kv"; done
  python bench.py --mode forward --steps 3 --warmup 1 --no-cpu-baseline $E 2>&1 | python -c "$P" "c3 forward, $o"
done
