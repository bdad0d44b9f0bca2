"""Developer tool (GPU): VJP step of a random circuit over all twelve gate kinds, with the lean lowerings of schedule.cpp
lower() (constant H / CNOT, Y, XX, YY powers) and without (QHBM_NO_LEAN_CLIFFORD=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import numpy as np, torch
from oracle import qhbm_oracle as O
from qhbmlib_amd import _engine as E
from tests.test_engine_gpu import random_circuit
n, n_gates, states = 18, 300, 128
rng = np.random.default_rng(0)
kinds = [int(k) for k in sys.argv[1].split(",")] if len(sys.argv) > 1 else None
gates = random_circuit(rng, n, n_gates, 8, kinds)
eng = E.Engine(0)
eng.set_circuit(n, gates, 8); eng.set_observables([O.xxz_chain_op(n)])
bits = torch.from_numpy(rng.integers(0, 2, size=(states, n)).astype(np.int8)).cuda()
params = torch.from_numpy(rng.uniform(-1, 1, 8).astype(np.float32)).cuda()
up = torch.full((states, 1), 1.0 / states, device="cuda")
eng.expectation_vjp(bits, params, up); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): eng.expectation_vjp(bits, params, up)
torch.cuda.synchronize()
print(f"kinds {sys.argv[1] if len(sys.argv) > 1 else 'all'} lean={'off' if os.environ.get('QHBM_NO_LEAN_CLIFFORD') else 'on'}: {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms per {states}-state step, passes {eng.num_passes()}", flush=True)
