"""Pins the numpy oracle to the closed-form known answers held by the
reference's own test-suite.  CPU only.

Each test cites the reference test (file:line under /root/reference) whose
expected values it re-states; the expected side is a closed form, so nothing
from the reference is read at run time.
"""
import itertools
import math

import numpy as np
import pytest

from oracle import qhbm_oracle as O


# ---- tests/inference/qnn_test.py:83-180 : X**p on 3 qubits -----------------
@pytest.mark.parametrize("p", [0.37, -0.81, 0.0, 1.0, 0.123])
def test_x_pow_expectations_and_grads(p):
  n = 3
  gates = [(O.GATE_XPOW, q, -1, 0, 1.0, 0.0) for q in range(n)]
  params = np.array([p])
  bits = np.array(5 * list(itertools.product([0, 1], repeat=n)), dtype=np.int8)
  sin, cos = math.sin(math.pi * p), math.cos(math.pi * p)
  for pauli, val, grad in (
      ("X", lambda s: 0.0, lambda s: 0.0),
      ("Y", lambda s: -((-1.0)**s) * sin, lambda s: -((-1.0)**s) * math.pi * cos),
      ("Z", lambda s: ((-1.0)**s) * cos, lambda s: -((-1.0)**s) * math.pi * sin),
  ):
    ops = [[O.pauli_term(1.0, [(q, pauli)])] for q in range(n)]
    vals, jac = O.expectation_jacobian(n, gates, params, bits, ops)
    exp_vals = np.array([[val(s) for s in row] for row in bits])
    exp_grad = np.array([[grad(s) for s in row] for row in bits])
    # reference tolerance: atol 2e-3 (qnn_test.py:49); the oracle is exact.
    np.testing.assert_allclose(vals, exp_vals, atol=1e-12)
    np.testing.assert_allclose(jac[:, :, 0], exp_grad, atol=1e-11)
    np.testing.assert_allclose(
        O.expectation(n, gates, params, bits, ops), exp_vals, atol=1e-12)


# ---- tests/inference/vqt_loss_test.py:133-205 : rx per qubit, H = sum Y ----
@pytest.mark.parametrize("n", [1, 2, 3])
def test_vqt_rx_closed_form(n):
  rng = np.random.default_rng(7 + n)
  thetas = rng.uniform(-2, 2, n)
  phis = rng.uniform(-1, 1, n)
  beta = rng.uniform(0.01, 3.0)
  # cirq.rx(phi) = XPowGate(exponent=phi/pi, global_shift=-0.5)
  gates = [(O.GATE_XPOW, q, -1, q, 1.0 / math.pi, 0.0) for q in range(n)]
  h_op = [O.pauli_term(1.0, [(q, "Y")]) for q in range(n)]
  # Every bitstring with its exact Bernoulli probability as weight replaces the
  # 1e7 Monte-Carlo samples of the reference test.
  all_bits = O.all_bitstrings(n)
  energies = O.bernoulli_energy(all_bits, thetas)
  probs = np.exp(-energies)
  probs /= probs.sum()
  vals = O.expectation(n, gates, phis, all_bits, [h_op])[:, 0]
  # NOTE the reference's closed form uses <0|rx^dag Y rx|0> = -sin(phi) and
  # p(1) = e^theta/(e^theta+e^-theta), giving sum tanh(theta) sin(phi).
  expectation = float(probs @ vals)
  np.testing.assert_allclose(
      expectation, np.sum(np.tanh(thetas) * np.sin(phis)), atol=1e-12)
  entropy = O.entropy_exact(lambda b: O.bernoulli_energy(b, thetas), n)
  np.testing.assert_allclose(
      entropy,
      np.sum(-thetas * np.tanh(thetas) + np.log(2 * np.cosh(thetas))),
      atol=1e-12)
  # loss = beta <H> - S ; with exact weights the sample average is exact.
  log_z = O.log_partition_exact(lambda b: O.bernoulli_energy(b, thetas), n)
  f = beta * vals - energies
  loss = float(probs @ f) - log_z
  np.testing.assert_allclose(loss, beta * expectation - entropy, atol=1e-12)
  # gradients (vqt_loss_test.py:193-197) through ebm.py:303-324 formulas.
  _, jac = O.expectation_jacobian(n, gates, phis, all_bits, [h_op])
  dphi = beta * (probs @ jac[:, 0, :])
  np.testing.assert_allclose(
      dphi, beta * np.tanh(thetas) * np.cos(phis), atol=1e-11)
  e_grads = O.spins_from_bitstrings(all_bits)
  dtheta = (probs @ e_grads) * (probs @ f) - probs @ (e_grads * f[:, None])
  np.testing.assert_allclose(
      dtheta, (1 - np.tanh(thetas)**2) * (beta * np.sin(phis) + thetas),
      atol=1e-11)


def test_vqt_loss_and_grads_helper_matches_weighted_form():
  """vqt_loss_and_grads on a multiset == the weighted closed form."""
  n = 2
  thetas = np.array([0.3, -0.7])
  phis = np.array([0.4, -0.2])
  beta = 1.3
  gates = [(O.GATE_XPOW, q, -1, q, 1.0 / math.pi, 0.0) for q in range(n)]
  h_op = [O.pauli_term(1.0, [(q, "Y")]) for q in range(n)]
  samples = np.array([[0, 0], [1, 0], [0, 0], [1, 1], [1, 0], [0, 0]], np.int8)
  log_z = float(np.sum(np.log(2 * np.cosh(thetas))))
  loss, dth, dph = O.vqt_loss_and_grads(
      n, gates, phis, samples, h_op, beta,
      lambda b: O.bernoulli_energy(b, thetas),
      O.spins_from_bitstrings, log_z)
  uniq, _, counts = O.unique_bitstrings_with_counts(samples)
  w = counts / counts.sum()
  vals, jac = O.expectation_jacobian(n, gates, phis, uniq, [h_op])
  f = beta * vals[:, 0] - O.bernoulli_energy(uniq, thetas)
  np.testing.assert_allclose(loss, w @ f - log_z, atol=1e-12)
  np.testing.assert_allclose(dph, beta * (w @ jac[:, 0, :]), atol=1e-12)
  s = O.spins_from_bitstrings(uniq)
  np.testing.assert_allclose(
      dth, (w @ s) * (w @ f) - w @ (s * f[:, None]), atol=1e-12)


# ---- tests/inference/qmhl_loss_test.py:136-272 : rx model vs ry data -------
@pytest.mark.parametrize("n", [1, 2, 3])
def test_qmhl_rx_ry_closed_form(n):
  rng = np.random.default_rng(31 + n)
  thetas = rng.uniform(0.25, 1.0, n)
  phis = rng.uniform(math.pi / 4, math.pi, n)
  alphas = rng.uniform(-math.pi, math.pi, n)
  data_probs = rng.uniform(0, 1, n)  # P(bit = 0)
  # total circuit = ry(alpha) data circuit + (rx(phi) model circuit)^-1;
  # params = [alphas..., phis...].
  data_gates = [(O.GATE_YPOW, q, -1, q, 1.0 / math.pi, 0.0) for q in range(n)]
  model_gates = [(O.GATE_XPOW, q, -1, n + q, 1.0 / math.pi, 0.0)
                 for q in range(n)]
  params = np.concatenate([alphas, phis])
  all_bits = O.all_bitstrings(n)
  # samples ~ Bernoulli(probs = 1 - data_probs): P(bit=1) = 1 - data_probs.
  w = np.prod(np.where(all_bits == 1, 1 - data_probs, data_probs), axis=1)
  shards = O.bernoulli_shards(n)
  total = data_gates + O.inverse_gates(model_gates)
  shard_vals, jac = O.expectation_jacobian(n, total, params, all_bits, shards)
  k_vals = shard_vals @ thetas
  expectation = float(w @ k_vals)
  np.testing.assert_allclose(
      expectation,
      np.sum(thetas * (2 * data_probs - 1) * np.cos(alphas) * np.cos(phis)),
      atol=1e-12)
  np.testing.assert_allclose(
      O.modular_hamiltonian_expectation(n, data_gates, model_gates, params,
                                        all_bits, shards, thetas)[:, 0],
      k_vals, atol=1e-12)
  # d/dtheta = averaged shard expectations; d/dphi through the inverse circuit.
  np.testing.assert_allclose(
      w @ shard_vals,
      (2 * data_probs - 1) * np.cos(alphas) * np.cos(phis), atol=1e-12)
  dphi = w @ np.einsum("bt,btp->bp", np.tile(thetas, (len(all_bits), 1)), jac)
  np.testing.assert_allclose(
      dphi[n:],
      -thetas * (2 * data_probs - 1) * np.cos(alphas) * np.sin(phis),
      atol=1e-11)
  log_z = O.log_partition_exact(lambda b: O.bernoulli_energy(b, thetas), n)
  np.testing.assert_allclose(
      log_z, np.sum(np.log(2 * np.cosh(thetas))), atol=1e-12)


# ---- self-VQT optimum: vqt_loss_test.py:46-83 ------------------------------
def test_self_vqt_is_minus_log_partition():
  """VQT of a QHBM against its own modular Hamiltonian at beta=1 is -log Z."""
  n = 3
  rng = np.random.default_rng(5)
  gates, names = O.hea_gates(n, 2, "v")
  params = rng.uniform(-1, 1, len(names))
  ix = O.parity_indices(n, n)
  thetas = rng.uniform(-1, 1, len(ix))
  all_bits = O.all_bitstrings(n)
  energies = O.kobe_energy(all_bits, thetas, n)
  probs = np.exp(-energies)
  probs /= probs.sum()
  k_vals = O.modular_hamiltonian_expectation(
      n, gates, gates, params, all_bits, O.kobe_shards(n, n), thetas)[:, 0]
  np.testing.assert_allclose(k_vals, energies, atol=1e-11)
  log_z = O.log_partition_exact(lambda b: O.kobe_energy(b, thetas, n), n)
  loss = float(probs @ (1.0 * k_vals - energies)) - log_z
  np.testing.assert_allclose(loss, -log_z, atol=1e-11)


# ---- tests/inference/qhbm_utils_test.py:28-51 : Bell state ------------------
def test_bell_state():
  gates = [(O.GATE_HPOW, 0, -1, -1, 0.0, 1.0),
           (O.GATE_CNOTPOW, 0, 1, -1, 0.0, 1.0)]
  psi = O.simulate(2, gates, [], [0, 0]).ravel()
  np.testing.assert_allclose(
      psi, np.array([1, 0, 0, 1]) / math.sqrt(2), atol=1e-12)
  # CNOT**1 and H**1 equal CNOT and H up to NO phase in cirq's convention.
  np.testing.assert_allclose(
      O.gate_matrix(O.GATE_CNOTPOW, 1.0),
      [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], atol=1e-12)


# ---- tests/models/energy_test.py:113-145, 233-249 --------------------------
def test_bernoulli_energy_simple():
  thetas = np.array([1.0, 1.7, -2.8])
  bits = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 1]])
  np.testing.assert_allclose(
      O.bernoulli_energy(bits, thetas),
      [thetas.sum(), -thetas[0] + thetas[1] + thetas[2],
       thetas[0] - thetas[1] - thetas[2]])


def test_kobe_energy_two_bits():
  np.testing.assert_allclose(
      O.kobe_energy([[0, 0], [0, 1], [1, 0], [1, 1]], [1.5, 2.7, -4.0], 2),
      [0.2, 2.8, 5.2, -8.2], atol=1e-12)


# ---- tests/inference/ebm_test.py:515-559 -----------------------------------
def test_kobe_log_partition_and_entropy():
  fn = lambda b: O.kobe_energy(b, [1.5, 2.7, -4.0], 2)
  np.testing.assert_allclose(
      O.log_partition_exact(fn, 2), math.log(3641.8353), rtol=1e-7)
  np.testing.assert_allclose(O.entropy_exact(fn, 2), 0.00233551808, rtol=2e-4)


# ---- tests/models/energy_utils_test.py:86-110 ------------------------------
def test_parity_indices_and_values():
  assert O.parity_indices(4, 3) == [
      (0,), (1,), (2,), (3,), (0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3),
      (0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3)]
  # spins [-1, 1, -1, -1] <-> bits [1, 0, 1, 1]
  np.testing.assert_array_equal(
      O.parities([[1, 0, 1, 1]], O.parity_indices(4, 3))[0],
      [-1, 1, -1, -1] + [-1, 1, 1, -1, -1, 1] + [1, 1, -1, 1])


# ---- tests/utils_test.py:47-73,107-186 -------------------------------------
def test_unique_first_occurrence_order_and_expand():
  bits = np.array([[1, 0], [0, 0], [1, 0], [1, 1], [0, 0], [1, 0]], np.int8)
  y, idx, counts = O.unique_bitstrings_with_counts(bits)
  np.testing.assert_array_equal(y, [[1, 0], [0, 0], [1, 1]])
  np.testing.assert_array_equal(idx, [0, 1, 0, 2, 1, 0])
  np.testing.assert_array_equal(counts, [3, 2, 1])
  np.testing.assert_array_equal(O.expand_unique_results(y, idx), bits)


def test_weighted_average():
  counts = np.array([1, 3])
  values = np.array([[2.0, 4.0], [6.0, 8.0]])
  np.testing.assert_allclose(O.weighted_average(counts, values), [5.0, 7.0])


# ---- gate conventions (cirq 0.14.1) ---------------------------
def test_gate_conventions():
  t = 0.37
  c, s = math.cos(math.pi * t / 2), math.sin(math.pi * t / 2)
  ph = np.exp(1j * math.pi * t / 2)
  np.testing.assert_allclose(
      O.gate_matrix(O.GATE_XPOW, t), ph * np.array([[c, -1j * s], [-1j * s, c]]),
      atol=1e-12)
  np.testing.assert_allclose(
      O.gate_matrix(O.GATE_YPOW, t), ph * np.array([[c, -s], [s, c]]),
      atol=1e-12)
  np.testing.assert_allclose(
      O.gate_matrix(O.GATE_ZPOW, t), np.diag([1, np.exp(1j * math.pi * t)]),
      atol=1e-12)
  np.testing.assert_allclose(
      O.gate_matrix(O.GATE_CZPOW, t),
      np.diag([1, 1, 1, np.exp(1j * math.pi * t)]), atol=1e-12)
  np.testing.assert_allclose(
      O.gate_matrix(O.GATE_ISWAPPOW, t),
      [[1, 0, 0, 0], [0, c, 1j * s, 0], [0, 1j * s, c, 0], [0, 0, 0, 1]],
      atol=1e-12)
  # rx(theta) = exp(-i theta X / 2) = XPOW(theta/pi) with global_shift -1/2.
  theta = 0.9
  np.testing.assert_allclose(
      O.gate_matrix(O.GATE_XPOW, theta / math.pi, -0.5),
      [[math.cos(theta / 2), -1j * math.sin(theta / 2)],
       [-1j * math.sin(theta / 2), math.cos(theta / 2)]], atol=1e-12)
  for kind in range(12):
    u = O.gate_matrix(kind, 0.61)
    np.testing.assert_allclose(u @ u.conj().T, np.eye(u.shape[0]), atol=1e-12)
    # X**1 == X etc.: the bit injector is exact (circuit.py:129-136).
  np.testing.assert_allclose(O.gate_matrix(O.GATE_XPOW, 1.0), O._X, atol=1e-12)
  np.testing.assert_allclose(O.gate_matrix(O.GATE_XPOW, 0.0), np.eye(2))


def test_every_gate_kind_against_the_documented_matrix():
  """All twelve kinds (and the rx / ry / rz forms) against cirq's documented closed-form matrices, written out
  entry by entry in tests/gate_docs.py -- not only unitarity (VERDICT r3 'finish the oracle pin')."""
  from tests import gate_docs as D
  for t in (0.37, -1.3, 1.0, 0.5):
    for kind in range(12):
      np.testing.assert_allclose(O.gate_matrix(kind, t), D.documented_matrix(kind, t), atol=1e-12, err_msg=f"kind {kind}")
  theta = 0.9
  for axis, kind in (("x", O.GATE_XPOW), ("y", O.GATE_YPOW), ("z", O.GATE_ZPOW)):
    np.testing.assert_allclose(O.gate_matrix(kind, theta / math.pi, -0.5), D.documented_rotation(axis, theta), atol=1e-12)
  # involutions at t = 1 (with cirq's phase: none for these kinds) and the sign convention of the controlled phase
  np.testing.assert_allclose(O.gate_matrix(O.GATE_HPOW, 1.0), O._H, atol=1e-12)
  np.testing.assert_allclose(O.gate_matrix(O.GATE_CNOTPOW, 1.0), [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], atol=1e-12)
  np.testing.assert_allclose(O.gate_matrix(O.GATE_SWAPPOW, 1.0), [[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], atol=1e-12)
  np.testing.assert_allclose(O.gate_matrix(O.GATE_ISWAPPOW, 1.0), [[1, 0, 0, 0], [0, 0, 1j, 0], [0, 1j, 0, 0], [0, 0, 0, 1]], atol=1e-12)


def _kat_gate(kind, q0, q1, t):
  return (kind, q0, q1, -1, 0.0, t)


def test_z_and_cz_powers_hand_derived_values_both_oracles():
  """Config 3's Z**t / CZ**t conventions made observable (no reference-held number covers them):
    H Z**t on |0>:            (|0> + e^{i pi t}|1>) / sqrt 2      =>  <X> = cos pi t, <Y> = sin pi t, <Z> = 0
    (H x H) CZ**t on |00>:    (|00> + |01> + |10> + e^{i pi t}|11>) / 2
                              =>  <X0 X1> = <X0> = <X1> = (1 + cos pi t) / 2,  <X0 Z1> = (1 - cos pi t) / 2,
                                  <Y0 Z1> = -sin(pi t) / 2 ... (sign: see below), <Z0 Z1> = 0
  derived by hand from the amplitudes; run through the numpy oracle and the C restatement."""
  from oracle import qhbm_cpu as C
  import os
  for t in (0.3, -0.85, 1.7):
    gates = [_kat_gate(O.GATE_HPOW, 0, -1, 1.0), _kat_gate(O.GATE_ZPOW, 0, -1, t)]
    ops = [[(1.0, 1, 0)], [(1.0, 1, 1)], [(1.0, 0, 1)]]   # X, Y, Z on qubit 0
    want = np.array([[math.cos(math.pi * t), math.sin(math.pi * t), 0.0]])
    bits = np.zeros((1, 1), np.int8)
    np.testing.assert_allclose(O.expectation(1, gates, [], bits, ops), want, atol=1e-12)
    if os.path.exists(C.LIB_PATH):
      np.testing.assert_allclose(C.expectation(1, gates, np.zeros(0, np.float32), bits, ops), want, atol=2e-6)
    gates = [_kat_gate(O.GATE_HPOW, 0, -1, 1.0), _kat_gate(O.GATE_HPOW, 1, -1, 1.0), _kat_gate(O.GATE_CZPOW, 0, 1, t)]
    cp, sp = math.cos(math.pi * t), math.sin(math.pi * t)
    # Y0 Z1: <psi|Y (x) Z|psi> with Y = [[0,-i],[i,0]]: pairs (|0b>, |1b>) with sign (-1)^b:
    #   b = 0: conj(a00)(-i)a10 + conj(a10)(i)a00 = 0 (both real, equal);  b = 1: -[conj(a01)(-i)a11 + conj(a11)(i)a01]
    #   = -(1/4)[-i e^{i pi t} + i e^{-i pi t}] = -(1/4)(2 sin pi t) = -sin(pi t) / 2
    ops = [[(1.0, 3, 0)], [(1.0, 1, 0)], [(1.0, 2, 0)], [(1.0, 1, 2)], [(1.0, 1, 3)], [(1.0, 0, 3)]]
    #       X0 X1          X0             X1             X0 Z1          Y0 Z1          Z0 Z1
    want = np.array([[(1 + cp) / 2, (1 + cp) / 2, (1 + cp) / 2, (1 - cp) / 2, -sp / 2, 0.0]])
    bits = np.zeros((1, 2), np.int8)
    np.testing.assert_allclose(O.expectation(2, gates, [], bits, ops), want, atol=1e-12)
    if os.path.exists(C.LIB_PATH):
      np.testing.assert_allclose(C.expectation(2, gates, np.zeros(0, np.float32), bits, ops), want, atol=2e-6)


def test_every_gate_kind_probe_values_from_documented_matrices_both_oracles():
  """X**p-prepared product states through each gate kind: expectation values computed from the documented matrices
  alone (tests/gate_docs.py::probe_values) against both oracles."""
  from oracle import qhbm_cpu as C
  from tests import gate_docs as D
  import os
  rng = np.random.default_rng(12)
  for kind in range(1, 12):
    nq = O.gate_num_qubits(kind)
    for _ in range(3):
      t = float(rng.uniform(-1.5, 1.5))
      probes = [float(rng.uniform(0.1, 0.9)) for _ in range(nq)]
      bits = rng.integers(0, 2, size=(1, nq)).astype(np.int8)
      gates = [_kat_gate(O.GATE_XPOW, q, -1, probes[q]) for q in range(nq)]
      gates.append(_kat_gate(kind, 0, 1 if nq == 2 else -1, t))
      strings = ["X", "Y", "Z"] if nq == 1 else ["XI", "IY", "ZZ", "XY", "YZ", "ZX", "YY"]
      code = {"I": (0, 0), "X": (1, 0), "Y": (1, 1), "Z": (0, 1)}
      ops = []
      for st in strings:
        x = sum(code[ch][0] << q for q, ch in enumerate(st))
        z = sum(code[ch][1] << q for q, ch in enumerate(st))
        ops.append([(1.0, x, z)])
      want = D.probe_values(kind, t, bits[0], probes, strings)[None, :]
      np.testing.assert_allclose(O.expectation(nq, gates, [], bits, ops), want, atol=1e-12, err_msg=f"kind {kind}")
      if os.path.exists(C.LIB_PATH):
        np.testing.assert_allclose(C.expectation(nq, gates, np.zeros(0, np.float32), bits, ops), want, atol=3e-6,
                                   err_msg=f"kind {kind} (C)")


def test_jacobian_matches_shift_rule_and_finite_differences():
  n = 3
  rng = np.random.default_rng(3)
  gates, names = O.hea_gates(n, 2, "t")
  params = rng.uniform(-1, 1, len(names))
  bits = O.all_bitstrings(n)
  ops = [O.tfim_ring_op(n), O.xxz_chain_op(n)]
  vals, jac = O.expectation_jacobian(n, gates, params, bits, ops)
  np.testing.assert_allclose(
      jac, O.expectation_parameter_shift(n, gates, params, bits, ops),
      atol=1e-10)
  eps = 1e-6
  for p in (0, 5, len(params) - 1):
    d = np.zeros_like(params)
    d[p] = eps
    fd = (O.expectation(n, gates, params + d, bits, ops) -
          O.expectation(n, gates, params - d, bits, ops)) / (2 * eps)
    np.testing.assert_allclose(jac[:, :, p], fd, atol=1e-7)
  np.testing.assert_allclose(vals, O.expectation(n, gates, params, bits, ops))


def test_tfq_bit_permutation():
  assert O.tfq_bit_permutation(4) == [0, 1, 2, 3]
  assert O.tfq_bit_permutation(12) == [0, 1, 10, 11, 2, 3, 4, 5, 6, 7, 8, 9]


def test_hea_structure():
  """tests/test_util_test.py:31-96 shapes: P = L(3n-1)."""
  for n, layers in ((4, 2), (12, 8), (5, 3)):
    gates, names = O.hea_gates(n, layers, "x")
    assert len(names) == layers * (3 * n - 1)
    assert len(gates) == len(names)
    assert sorted(g[3] for g in gates) == list(range(len(names)))


# ---- tests/inference/qhbm_utils_test.py:28-51 : density matrix of the Bell state --------------
def test_bell_density_matrix():
  gates = [(O.GATE_HPOW, 0, -1, -1, 0.0, 1.0),
           (O.GATE_CNOTPOW, 0, 1, -1, 0.0, 1.0)]
  thetas = np.array([-10.0, -10.0])  # pins the EBM at |00>
  dm = O.density_matrix(2, gates, [], lambda b: O.bernoulli_energy(b, thetas))
  expected = np.array([[0.5, 0, 0, 0.5], [0, 0, 0, 0], [0, 0, 0, 0], [0.5, 0, 0, 0.5]])
  np.testing.assert_allclose(dm, expected, atol=1e-6)


# ---- tests/inference/qhbm_utils_test.py:61-80 : fidelity of a model with itself is 1 ----------
def test_fidelity_self_is_one():
  n = 4
  gates, names = O.hea_gates(n, 3, "f")
  rng = np.random.default_rng(11)
  params = rng.uniform(-1, 1, len(names))
  thetas = rng.uniform(-1, 1, n)
  dm = O.density_matrix(n, gates, params, lambda b: O.bernoulli_energy(b, thetas))
  np.testing.assert_allclose(np.trace(dm).real, 1.0, atol=1e-12)
  np.testing.assert_allclose(O.fidelity_direct(dm, dm), 1.0, rtol=1e-6)
  u = O.unitary(n, gates, params)
  np.testing.assert_allclose(u.conj().T @ u, np.eye(2**n), atol=1e-12)
