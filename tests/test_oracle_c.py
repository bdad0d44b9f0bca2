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
  """A call that holds very few large states (here one of 22 qubits; from 26 qubits always: TFQ's policy for large
  circuits) splits every sweep over a team; the arithmetic per amplitude is the same, sums are double-precision reductions."""
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


@pytest.mark.parametrize("n", [5, 6, 9, 13])
def test_timed_baseline_path_equals_the_gate_by_gate_checker(n):
  """oracle/qhbm_cpu_diag.c ("port+diag": merged diagonal runs, AVX2 one-qubit kernel, fused adjoint steps -- what
  bench.py times as `cpu_baseline`) against oracle/qhbm_cpu.c on HEA circuits and on random circuits of every gate kind
  (diagonal runs cut by the straddling-pair limit, generic two-qubit gates, parameter-free gates): values 1e-6 * sum|c|,
  gradients 1e-5 relative, forward-only mode included."""
  rng = np.random.default_rng(50 + n)
  for trial in range(3):
    if trial == 0:
      gates, names = O.hea_gates(n, 3, "d")
      n_params = len(names)
    else:
      n_params = 7
      gates = _random_circuit(rng, n, 70, n_params)
      if trial == 2:   # long diagonal runs over scattered pairs: more than four straddling high bits
        for _ in range(40):
          q0 = int(rng.integers(n)); q1 = int(rng.integers(n - 1)); q1 += q1 >= q0
          kind = [3, 5, 11][int(rng.integers(3))]
          gates.append((kind, q0, q1 if kind != 3 else -1, int(rng.integers(n_params)), float(rng.uniform(-1, 1)), 0.1))
    params = rng.uniform(-1, 1, n_params)
    ops = [O.xxz_chain_op(n), O.tfim_ring_op(n), O.random_pauli_op(n, 6, trial, 0.5)]
    bits = rng.integers(0, 2, size=(4, n)).astype(np.int8)
    up = rng.normal(size=(4, 3))
    want_v, want_g = C.expectation_vjp(n, gates, params, bits, ops, up)
    got_v, got_g = C.expectation_vjp_diag(n, gates, params, bits, ops, up)
    fwd_v, none = C.expectation_vjp_diag(n, gates, params, bits, ops)
    norm = np.array([sum(abs(c) for c, _, _ in op) for op in ops])
    assert none is None
    assert (np.abs(got_v - want_v) <= 1e-6 * norm).all() and (np.abs(fwd_v - want_v) <= 1e-6 * norm).all()
    np.testing.assert_allclose(got_g, want_g, atol=1e-5 * max(1.0, np.abs(want_g).max()), rtol=0)
