"""Randomised parity stress of the block-grouped Pauli-sum kernels against the numpy oracle (developer tool; GPU):
qubit counts 13..18, 1..4 observables, sparse and dense strings, 1..11 states (whole groups of eight AND leftovers:
both XCD maps and both pivot rules of the value modes), values / VJP / retained backward.
    python scripts/experiments/stress_observable_blocks.py [seeds [first seed]]
(QHBM_OBS_BLOCK_BITS=12 in the environment: the same sweep on the kernel's second shape, option "observable_block_bits".)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qhbm-library_amd"))
import numpy as np
from oracle import qhbm_oracle as O
from tests.test_observable_blocks_gpu import _engine, _random_ops, _check

bad, t0 = 0, time.time()
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for seed in range(first, first + (int(sys.argv[1]) if len(sys.argv) > 1 else 40)):
  rng = np.random.default_rng(7000 + seed)
  n = 13 + seed % 6
  n_ops = 1 + seed % 4
  terms = [1, 7, 40, 150][(seed // 2) % 4]
  p_id = [0.5, 0.75, 0.9][seed % 3]
  states = [1, 3, 8, 9, 11][seed % 5] if n <= 16 else [1, 2, 3][seed % 3]
  gates, names = O.hea_gates(n, 1 + seed % 2, "s")
  params = rng.uniform(-1, 1, len(names))
  ops = _random_ops(rng, n, n_ops, terms, p_identity=p_id)
  bits = rng.integers(0, 2, size=(states, n)).astype(np.int8)
  up = rng.normal(size=(states, n_ops)).astype(np.float32)
  for opts in ({"observable_kernel": 1}, {"observable_kernel": 1, "observable_xcd_states": 0},
               {"observable_kernel": 1, "observable_xcd_states": 1, "multi_observable_values": 1}):
    try:
      _check(_engine(n, gates, len(names), ops, **opts), n, gates, params, bits, ops, up)
    except AssertionError as e:
      bad += 1
      print("FAIL seed", seed, "n", n, "ops", n_ops, "terms", terms, "states", states, opts, str(e)[:160], flush=True)
print(f"done: {bad} failures in {time.time() - t0:.0f} s")
