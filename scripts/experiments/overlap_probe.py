"""Developer probe: do a forward sweep and an adjoint sweep of two different chunks overlap usefully when they
run on two HIP streams?  Two engines (own workspaces), config-3 circuit, `states` states each; stream 2 is
offset by one forward-only call so that its forward passes meet stream 1's adjoint passes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import numpy as np, torch
import bench
from qhbmlib_amd import _engine as E
n, layers, states = 20, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = 6
gates, P = bench.hea_gates(n, layers)
engs = []
for _ in range(2):
  e = E.Engine(0); e.set_circuit(n, gates, P); e.set_observables([bench.xxz_op(n)]); engs.append(e)
bits = [torch.from_numpy(bench.distinct_bitstrings(n, states, 1 + i)).cuda() for i in range(2)]
params = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, P).astype(np.float32)).cuda()
up = torch.full((states, 1), 1.0 / states, device="cuda")
for e, b in zip(engs, bits): e.expectation_vjp(b, params, up)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
  for e, b in zip(engs, bits): e.expectation_vjp(b, params, up)
torch.cuda.synchronize()
seq = (time.perf_counter() - t0) / (2 * reps)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(s2):
  engs[1].expectation(bits[1], params)  # offset: one forward
for _ in range(reps):
  with torch.cuda.stream(s1): engs[0].expectation_vjp(bits[0], params, up)
  with torch.cuda.stream(s2): engs[1].expectation_vjp(bits[1], params, up)
torch.cuda.synchronize()
par = (time.perf_counter() - t0)
fwd_only = None
torch.cuda.synchronize(); t1 = time.perf_counter(); engs[1].expectation(bits[1], params); torch.cuda.synchronize(); fwd_only = time.perf_counter() - t1
print(f"states/call {states}: sequential {seq*1e3:.2f} ms per call; two streams {(par - fwd_only) / (2 * reps) * 1e3:.2f} ms per call (offset forward {fwd_only*1e3:.2f} ms subtracted)")
