"""A/B timing of engine options on the C3 shard (developer tool): python scripts/ab.py states opt=val,opt=val ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import numpy as np, torch
import bench
from qhbmlib_amd import _engine as E
n, layers = 20, 16
states = int(sys.argv[1])
gates, P = bench.hea_gates(n, layers)
op = bench.xxz_op(n)
bits = torch.from_numpy(bench.distinct_bitstrings(n, states, 1)).cuda()
params = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, P).astype(np.float32)).cuda()
up = torch.full((states, 1), 1.0 / states, device="cuda")
engs = []
for spec in sys.argv[2:]:
  eng = E.Engine(0)
  for kv in spec.split(","):
    if kv and kv != "default":
      k, v = kv.split("="); eng.set_option(k, int(v))
  eng.set_circuit(n, gates, P); eng.set_observables([op])
  eng.expectation_vjp(bits, params, up)
  engs.append((spec, eng))
torch.cuda.synchronize()
res = {s: [[], []] for s, _ in engs}
for rep in range(4):
  for spec, eng in engs:
    torch.cuda.synchronize(); t0 = time.perf_counter(); eng.expectation(bits, params); torch.cuda.synchronize()
    res[spec][0].append(time.perf_counter() - t0)
    t0 = time.perf_counter(); eng.expectation_vjp(bits, params, up); torch.cuda.synchronize()
    res[spec][1].append(time.perf_counter() - t0)
for spec, (f, v) in res.items():
  print(f"{spec:60s} fwd {min(f)/states*1e6:8.2f} us/state   vqt {min(v)/states*1e6:8.2f} us/state", flush=True)
