"""Per-pass kernel times of one VQT step on the C3 shard (developer tool): HIP events around every launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import numpy as np, torch
import bench
from qhbmlib_amd import _engine as E
n, layers, states = 20, 16, int(sys.argv[1])
gates, P = bench.hea_gates(n, layers)
eng = E.Engine(0)
for kv in filter(None, os.environ.get("QHBM_OPTS", "").split(",")):
  eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
eng.set_circuit(n, gates, P); eng.set_observables([bench.xxz_op(n)])
bits = torch.from_numpy(bench.distinct_bitstrings(n, states, 1)).cuda()
params = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, P).astype(np.float32)).cuda()
up = torch.full((states, 1), 1.0 / states, device="cuda")
eng.set_option("profile_events", 1)
eng.expectation_vjp(bits, params, up); torch.cuda.synchronize(); eng.kernel_time_ms(True)
eng.expectation_vjp(bits, params, up); torch.cuda.synchronize(); kt = eng.kernel_time_ms(True)
print(f"{os.path.basename(E.LIB_PATH):24s} {states} states: forward {kt['fwd_ms']:.2f} ms ({kt['fwd_launches']}), lambda {kt['obs_ms']:.2f}, adjoint {kt['bwd_ms']:.2f} ms ({kt['bwd_launches']})")
