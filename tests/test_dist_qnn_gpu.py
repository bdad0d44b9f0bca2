"""The host mirror's sharded expectation on the GPU: `AnalyticQuantumInference` with a
process group splits the unique bitstrings over the ranks, all-gathers the values and adds the
per-state gradient rows in global state order.  Results must be BIT-IDENTICAL for 1, 2 and 3 ranks
(3 does not divide the 11 unique states) and agree with the oracle."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, out_path, backend="gloo", ordered=True):
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_qnn_worker.py"),
         out_path]
  out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, QHBM_TEST_BACKEND=backend, QHBM_TEST_ORDERED="1" if ordered else "0"))
  assert out.returncode == 0, out.stderr[-3000:]
  return dict(np.load(out_path))


def test_sharded_expectation_is_bit_identical_for_any_number_of_ranks(tmp_path):
  runs = [_run(w, str(tmp_path / f"w{w}.npz")) for w in (1, 2, 3)]
  assert [int(r["world"]) for r in runs] == [1, 2, 3]
  for r in runs[1:]:
    np.testing.assert_array_equal(r["values"], runs[0]["values"])     # atol = 0
    np.testing.assert_array_equal(r["grad"], runs[0]["grad"])
  # the RCCL code path (device tensors straight into the collectives, no host staging): one rank is all
  # a one-GPU box can run over nccl, and it must give the same bits
  rccl = _run(1, str(tmp_path / "rccl.npz"), backend="nccl")
  np.testing.assert_array_equal(rccl["values"], runs[0]["values"])
  np.testing.assert_array_equal(rccl["grad"], runs[0]["grad"])
  # the DEFAULT reduction (one all-reduce of the [P] gradient instead of the gathered rows): the same numbers up to
  # the order of the fp32 sum, values identical (they are gathered either way)
  for w in (1, 2):
    ar = _run(w, str(tmp_path / f"allreduce{w}.npz"), ordered=False)
    np.testing.assert_array_equal(ar["values"], runs[0]["values"])
    np.testing.assert_allclose(ar["grad"], runs[0]["grad"], rtol=0, atol=2e-6 * max(1.0, np.abs(runs[0]["grad"]).max()))
  # and they are right: the numpy oracle on the same model
  from oracle import qhbm_oracle as O
  n, layers = 14, 2
  rng = np.random.default_rng(14)
  gates, names = O.hea_gates(n, layers, "d")
  params = rng.uniform(-1, 1, len(names)).astype(np.float32).astype(np.float64)
  ops = [O.xxz_chain_op(n), [O.pauli_term(1.0, [(q, "Z")]) for q in range(n)]]
  uniq = rng.integers(0, 2, size=(11, n)).astype(np.int8)
  states = np.concatenate([uniq, uniq[[3, 3, 7]]])
  weights = rng.normal(size=(states.shape[0], 2)).astype(np.float32)
  want, jac = O.expectation_jacobian(n, gates, params, states, ops)
  np.testing.assert_allclose(runs[0]["values"], want, atol=2e-5 * 19)
  want_grad = np.einsum("bt,btp->p", weights.astype(np.float64), jac)
  np.testing.assert_allclose(runs[0]["grad"], want_grad, atol=1e-4 * max(1.0, np.abs(want_grad).max()))
