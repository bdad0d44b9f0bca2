"""VQT-step timing on the C3 shard with the library named by QHBM_ENGINE_LIB (developer tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import numpy as np, torch
import bench
from qhbmlib_amd import _engine as E
n, layers, states = 20, 16, int(sys.argv[1])
gates, P = bench.hea_gates(n, layers)
eng = E.Engine(0)
for kv in filter(None, os.environ.get("QHBM_OPTS", "").split(",")):   # engine options for A/B runs: QHBM_OPTS=name=value,...
  eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
eng.set_circuit(n, gates, P); eng.set_observables([bench.xxz_op(n)])
bits = torch.from_numpy(bench.distinct_bitstrings(n, states, 1)).cuda()
params = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, P).astype(np.float32)).cuda()
up = torch.full((states, 1), 1.0 / states, device="cuda")
eng.set_option("profile_events", 1)
eng.expectation_vjp(bits, params, up); torch.cuda.synchronize(); eng.kernel_time_ms(True)
for _ in range(3): eng.expectation_vjp(bits, params, up)
torch.cuda.synchronize(); kt = eng.kernel_time_ms(True)
print(f"{os.path.basename(E.LIB_PATH):24s} {os.environ.get('QHBM_OPTS', ''):28s} adjoint passes {kt['bwd_ms']/3/states*1e3:8.2f} us/state   "
      f"forward {kt['fwd_ms']/3/states*1e3:8.2f}   lambda = O psi {kt['obs_ms']/3/states*1e3:8.2f}", flush=True)
