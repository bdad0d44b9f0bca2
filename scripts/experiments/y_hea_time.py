"""Developer tool (GPU): config 3's HEA with Y**t in place of X**t (ry-style ansatz): step time of the engine's VJP call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import numpy as np, torch
import bench
from qhbmlib_amd import _engine as E
n, layers, states = 20, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 512
gates, P = bench.hea_gates(n, layers)
for kind_name in ("X", "Y"):
  gs = [((E.GATE_YPOW if kind_name == "Y" and g[0] == E.GATE_XPOW else g[0]),) + tuple(g[1:]) for g in gates]
  eng = E.Engine(0)
  eng.set_circuit(n, gs, P); eng.set_observables([bench.xxz_op(n)])
  bits = torch.from_numpy(bench.distinct_bitstrings(n, states, 1)).cuda()
  params = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, P).astype(np.float32)).cuda()
  up = torch.full((states, 1), 1.0 / states, device="cuda")
  eng.expectation_vjp(bits, params, up); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(3): eng.expectation_vjp(bits, params, up)
  torch.cuda.synchronize()
  print(f"{kind_name}-HEA: {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms per {states}-state VQT step, passes {eng.num_passes()}", flush=True)
