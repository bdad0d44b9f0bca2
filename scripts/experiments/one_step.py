"""Runs a single forward / VQT step (for rocprofv3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import numpy as np, torch
import bench
from qhbmlib_amd import _engine as E
n, layers, states, ham, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
opts = dict(kv.split("=") for kv in sys.argv[6:])
gates, P = bench.hea_gates(n, layers)
op = bench.xxz_op(n) if ham == "xxz" else bench.tfim_op(n)
eng = E.Engine(0)
for k, v in opts.items(): eng.set_option(k, int(v))
eng.set_circuit(n, gates, P); eng.set_observables([op])
bits = torch.from_numpy(bench.distinct_bitstrings(n, states, 1)).cuda()
params = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, P).astype(np.float32)).cuda()
up = torch.full((states, 1), 1.0 / states, device="cuda")
for _ in range(2):
  if mode == "fwd": eng.expectation(bits, params)
  else: eng.expectation_vjp(bits, params, up)
torch.cuda.synchronize()
