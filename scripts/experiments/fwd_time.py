"""Forward-only timing on the C3 shard with the library named by QHBM_ENGINE_LIB (developer tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import numpy as np, torch
import bench
from qhbmlib_amd import _engine as E
n, layers, states = 20, 16, int(sys.argv[1])
gates, P = bench.hea_gates(n, layers)
eng = E.Engine(0); eng.set_circuit(n, gates, P); eng.set_observables([bench.xxz_op(n)])
bits = torch.from_numpy(bench.distinct_bitstrings(n, states, 1)).cuda()
params = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, P).astype(np.float32)).cuda()
eng.expectation(bits, params); torch.cuda.synchronize()
ts = []
for _ in range(4):
  t0 = time.perf_counter(); eng.expectation(bits, params); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"{os.path.basename(E.LIB_PATH):24s} fwd {min(ts)/states*1e6:8.2f} us/state", flush=True)
