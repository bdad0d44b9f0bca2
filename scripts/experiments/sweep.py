"""Tuning sweep (developer tool): times forward / VQT step for tile and round geometries."""
import itertools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import numpy as np, torch
import bench
from qhbmlib_amd import _engine as E

def run(n, layers, states, ham, tile, rnd, adj_tile, mode, reps=3):
  gates, P = bench.hea_gates(n, layers)
  op = bench.xxz_op(n) if ham == "xxz" else bench.tfim_op(n)
  eng = E.Engine(0)
  if tile: eng.set_option("tile_qubits", tile)
  if rnd: eng.set_option("round_qubits", rnd)
  if adj_tile: eng.set_option("adjoint_tile_qubits", adj_tile)
  eng.set_circuit(n, gates, P); eng.set_observables([op])
  bits = torch.from_numpy(bench.distinct_bitstrings(n, states, 1)).cuda()
  params = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, P).astype(np.float32)).cuda()
  up = torch.full((states, 1), 1.0 / states, device="cuda")
  f = (lambda: eng.expectation(bits, params)) if mode == "fwd" else (lambda: eng.expectation_vjp(bits, params, up))
  f(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps): f()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / reps
  return dt, eng.num_passes()

if __name__ == "__main__":
  n, layers, states, ham = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
  for spec in sys.argv[5:]:
    mode, tile, rnd, adj = spec.split(",")
    try:
      dt, passes = run(n, layers, states, ham, int(tile), int(rnd), int(adj), mode)
      print(f"{mode} tile={tile} round={rnd} adj_tile={adj}: {dt*1e3:9.3f} ms  ({dt/states*1e6:8.2f} us/state) passes={passes}", flush=True)
    except Exception as exc:
      print(f"{mode} tile={tile} round={rnd} adj_tile={adj}: FAILED {exc}", flush=True)
