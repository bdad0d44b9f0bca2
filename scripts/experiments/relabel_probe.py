"""Developer probe: the config-3 step with the chain's qubits relabelled so that a chosen block of four
consecutive chain sites sits on the four lowest index bits (which every tile holds): how many passes and
how long?   python scripts/relabel_probe.py [states] [anchor ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import numpy as np, torch
import bench
from qhbmlib_amd import _engine as E
n, layers = 20, 16
states = int(sys.argv[1]) if len(sys.argv) > 1 else 512
anchors = [int(a) for a in sys.argv[2:]] or [16, 8]
gates0, P = bench.hea_gates(n, layers)
for a in anchors:
  # chain site s -> engine qubit label; label l sits on index bit n-1-l
  order = list(range(a, a + 4)) + [s for s in range(n) if not a <= s < a + 4]   # sites by index bit 0, 1, ...
  label = {s: n - 1 - b for b, s in enumerate(order)}
  gates = [(k, label[q0], label[q1] if q1 >= 0 else -1, p, sc, off) for (k, q0, q1, p, sc, off) in gates0]
  terms = []
  for i in range(n - 1):
    qa, qb = label[i], label[i + 1]
    m = (1 << qa) | (1 << qb)
    terms += [(1.0, m, 0), (1.0, m, m), (0.5, 0, m)]
  eng = E.Engine(0); eng.set_circuit(n, gates, P); eng.set_observables([terms])
  bits = torch.from_numpy(bench.distinct_bitstrings(n, states, 1)).cuda()
  params = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, P).astype(np.float32)).cuda()
  up = torch.full((states, 1), 1.0 / states, device="cuda")
  eng.set_option("profile_events", 1)
  eng.expectation_vjp(bits, params, up); torch.cuda.synchronize(); eng.kernel_time_ms(True)
  for _ in range(3): eng.expectation_vjp(bits, params, up)
  torch.cuda.synchronize(); kt = eng.kernel_time_ms(True)
  print(f"anchor sites {a}..{a+3}: passes {eng.num_passes()}  forward {kt['fwd_ms']/3/states*1e3:7.2f}  obs {kt['obs_ms']/3/states*1e3:6.2f}  adjoint {kt['bwd_ms']/3/states*1e3:7.2f} us/state", flush=True)
