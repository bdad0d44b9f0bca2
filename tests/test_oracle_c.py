"""The C restatement (oracle/qhbm_cpu.c, fp32) against the numpy oracle (complex128)."""
import os

import numpy as np
import pytest

from oracle import qhbm_oracle as O
from oracle import qhbm_cpu as C

pytestmark = pytest.mark.skipif(not os.path.exists(C.LIB_PATH),
                                reason="oracle/libqhbm_cpu.so not built (run __graft_entry__.build())")


def _random_circuit(rng, n, n_gates, n_params):
  gates = []
  for _ in range(n_gates):
    kind = int(rng.integers(12))
    q0 = int(rng.integers(n))
    q1 = -1
    if O.gate_num_qubits(kind) == 2:
      q1 = int(rng.integers(n - 1))
      q1 += q1 >= q0
    pidx = int(rng.integers(n_params)) if rng.random() < 0.8 else -1
    gates.append((kind, q0, q1, pidx, float(rng.uniform(-1.5, 1.5)) if pidx >= 0 else 0.0,
                  float(rng.uniform(-0.5, 0.5))))
  return gates


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_c_matches_numpy_all_kinds(seed):
  n, n_params = 5, 6
  rng = np.random.default_rng(seed)
  gates = _random_circuit(rng, n, 40, n_params)
  params = rng.uniform(-1, 1, n_params)
  ops = [O.random_pauli_op(n, 6, seed, 0.4), O.tfim_ring_op(n), O.xxz_chain_op(n)] + O.kobe_shards(n, 2)[:3]
  bits = O.all_bitstrings(n)
  want, jac = O.expectation_jacobian(n, gates, params, bits, ops)
  got = C.expectation(n, gates, params, bits, ops)
  np.testing.assert_allclose(got, want, atol=2e-5)
  up = rng.normal(size=want.shape)
  vals, grad = C.expectation_vjp(n, gates, params, bits, ops, up)
  np.testing.assert_allclose(vals, want, atol=2e-5)
  want_grad = np.einsum("bt,btp->p", up, jac)
  np.testing.assert_allclose(grad, want_grad, atol=2e-4 * max(1, np.abs(want_grad).max()))


def test_c_hea_tfim_n10():
  n = 10
  rng = np.random.default_rng(3)
  gates, names = O.hea_gates(n, 3, "c")
  params = rng.uniform(-1, 1, len(names))
  bits = rng.integers(0, 2, size=(3, n)).astype(np.int8)
  ops = [O.tfim_ring_op(n)]
  np.testing.assert_allclose(C.expectation(n, gates, params, bits, ops),
                             O.expectation(n, gates, params, bits, ops), atol=1e-4)


def test_c_statevector_matches_numpy():
  n = 8
  rng = np.random.default_rng(5)
  gates, names = O.hea_gates(n, 2, "sv")
  params = rng.uniform(-1, 1, len(names))
  bits = rng.integers(0, 2, size=(4, n)).astype(np.int8)
  want = np.stack([O.simulate(n, gates, params, b).reshape(-1) for b in bits])
  np.testing.assert_allclose(C.statevector(n, gates, params, bits), want, atol=2e-6)


def test_c_threads_inside_a_state_equal_the_serial_sweep():
  """From 18 qubits a call with fewer states than threads splits every sweep over the team (from 26 always: TFQ's
  policy for large circuits); the arithmetic per amplitude is the same, sums are double-precision reductions."""
  team = C.max_threads()
  if team < 4:
    pytest.skip("needs a team of at least four threads")
  n, n_params = 22, 5
  rng = np.random.default_rng(8)
  gates = _random_circuit(rng, n, 14, n_params)
  params = rng.uniform(-1, 1, n_params)
  ops = [O.random_pauli_op(n, 4, 1, 0.8), [(0.5, 0, 3 << 7), (0.25, 1 << 20, 0)]]
  bits = rng.integers(0, 2, size=(1, n)).astype(np.int8)
  up = rng.normal(size=(1, 2))
  v_team, g_team = C.expectation_vjp(n, gates, params, bits, ops, up)
  v_one, g_one = C.expectation_vjp(n, gates, params, bits, ops, up, n_threads=1)
  sv_team = C.statevector(n, gates, params, bits, n_threads=team)
  sv_one = C.statevector(n, gates, params, bits, n_threads=1)
  C.expectation(4, [], params, bits[:, :4], [[(1.0, 0, 1)]], n_threads=team)  # (the team size is sticky: restore it)
  assert C.max_threads() == team
  np.testing.assert_allclose(v_team, v_one, atol=1e-6)
  np.testing.assert_allclose(g_team, g_one, atol=1e-5 * max(1.0, np.abs(g_one).max()))
  np.testing.assert_allclose(sv_team, sv_one, atol=1e-7)
