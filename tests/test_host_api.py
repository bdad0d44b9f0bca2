"""Host-side mirror of the reference's operator interface: CPU-only checks
(IR, circuit algebra, energies, unique/expand, error behaviour).  The expected
values are the reference's own (file:line cited per test)."""
import itertools
import math

import numpy as np
import pytest
import torch

from oracle import qhbm_oracle as O
from qhbmlib_amd import ir, models, utils
from qhbmlib_amd.models import circuit_utils


def hea_circuit(qubits, num_layers, name):
  """tests/test_util.py:25-67 in the host IR."""
  circuit = ir.Circuit()
  for layer in range(num_layers):
    for n, q in enumerate(qubits):
      sx, sz = ir.symbols(f"sx_{name}_{layer}_{n} sz_{name}_{layer}_{n}")
      circuit += [ir.X(q)**sx, ir.Z(q)**sz]
    if len(qubits) > 1:
      for n, (q0, q1) in enumerate(zip(qubits[::2], qubits[1::2])):
        circuit += ir.CZPowGate(ir.Symbol(f"sc_{name}_{layer}_{2 * n}"))(q0, q1)
      shifted = qubits[1:]
      for n, (q0, q1) in enumerate(zip(shifted[::2], shifted[1::2])):
        circuit += ir.CZPowGate(ir.Symbol(f"sc_{name}_{layer}_{2 * n + 1}"))(q0, q1)
  return circuit


@pytest.mark.parametrize("n,layers", [(1, 2), (2, 1), (4, 2), (5, 3), (12, 2)])
def test_hea_ir_matches_oracle_flat_gates(n, layers):
  """The IR's lowering equals the oracle's independent construction, including the
  lexicographic variable layout."""
  qubits = ir.GridQubit.rect(1, n)
  circ = models.DirectQuantumCircuit(hea_circuit(qubits, layers, "m"))
  want_gates, want_names = O.hea_gates(n, layers, "m")
  assert circ.symbol_names == want_names
  got = circ.pqc.flat_gates(circ.qubits, circ.symbol_names)
  assert got == [tuple(g) for g in want_gates]
  assert circ.symbol_values.shape == (layers * (3 * n - 1),)


def test_circuit_add_and_inverse():
  """circuit.py:138-178, tests/models/circuit_test.py:161-231."""
  qubits = ir.GridQubit.rect(1, 3)
  a = models.DirectQuantumCircuit(ir.Circuit(ir.X(q)**ir.Symbol(f"a{i}") for i, q in enumerate(qubits)),
                                  name="a")
  b = models.DirectQuantumCircuit(ir.Circuit(ir.Y(q)**(2.0 * ir.Symbol(f"b{i}")) for i, q in enumerate(qubits)),
                                  name="b")
  total = a + b
  assert total.symbol_names == a.symbol_names + b.symbol_names
  assert total.name == "a_b"
  assert len(total.pqc) == 6
  assert [id(p) for p in total.trainable_variables] == [id(p) for p in a.trainable_variables + b.trainable_variables]
  inv = total**-1
  assert inv.name == "a_b_inverse"
  assert inv.symbol_names == total.symbol_names
  gates = inv.pqc.gates
  assert gates[0].kind == b.pqc.gates[-1].kind and gates[0].exponent.scalar == -2.0
  assert gates[-1].exponent.scalar == -1.0
  # same variables
  assert torch.equal(inv.symbol_values, total.symbol_values)
  with pytest.raises(ValueError, match="symbols in common"):
    _ = a + a
  with pytest.raises(ValueError):
    _ = a**2
  with pytest.raises(TypeError):
    _ = a + 1


def test_bit_injection_order_quirk():
  """circuit.py:59-62,132-134."""
  assert circuit_utils.tfq_bit_permutation(4) == [0, 1, 2, 3]
  assert circuit_utils.tfq_bit_permutation(12) == O.tfq_bit_permutation(12)
  qubits = ir.GridQubit.rect(1, 12)
  pqc = ir.Circuit(ir.X(q)**ir.Symbol(f"s{i}") for i, q in enumerate(qubits))
  assert models.DirectQuantumCircuit(pqc).bit_column_to_qubit() == list(range(12))
  assert models.DirectQuantumCircuit(pqc, tfq_compat_bit_order=True).bit_column_to_qubit()[:4] == [0, 1, 10, 11]


def test_unchosen_bit_order_is_announced_once_from_eleven_qubits_on():
  """From 11 qubits on the reference's lexicographic symbol sort permutes the bitstring columns
  (circuit.py:59-62,131-134): the mirror's default order then differs from the reference on the same inputs, and a
  circuit built without choosing says so -- once per process; an explicit True / False is silent."""
  import warnings
  from qhbmlib_amd.models import circuit as circuit_module
  pqc12 = ir.Circuit(ir.X(q)**ir.Symbol(f"s{i}") for i, q in enumerate(ir.GridQubit.rect(1, 12)))
  pqc10 = ir.Circuit(ir.X(q)**ir.Symbol(f"s{i}") for i, q in enumerate(ir.GridQubit.rect(1, 10)))
  circuit_module._bit_order_warned = False
  with warnings.catch_warnings(record=True) as seen:
    warnings.simplefilter("always")
    models.DirectQuantumCircuit(pqc10)                                # below 11 qubits the orders coincide
    models.DirectQuantumCircuit(pqc12, tfq_compat_bit_order=False)    # chosen
    models.DirectQuantumCircuit(pqc12, tfq_compat_bit_order=True)
    assert not [w for w in seen if issubclass(w.category, circuit_module.BitOrderWarning)]
    c = models.DirectQuantumCircuit(pqc12)
    assert c.tfq_compat_bit_order is False
    _ = c + models.DirectQuantumCircuit(ir.Circuit(ir.Z(q)**ir.Symbol(f"t{i}") for i, q in enumerate(ir.GridQubit.rect(1, 12))))
    _ = c**-1
    hits = [w for w in seen if issubclass(w.category, circuit_module.BitOrderWarning)]
    assert len(hits) == 1 and "tfq_compat_bit_order=True" in str(hits[0].message)


def test_qubits_sorted_row_major():
  qs = [ir.GridQubit(1, 0), ir.GridQubit(0, 1), ir.GridQubit(0, 0)]
  c = models.DirectQuantumCircuit(ir.Circuit(ir.X(q)**ir.Symbol(f"s{i}") for i, q in enumerate(qs)))
  assert c.qubits == [ir.GridQubit(0, 0), ir.GridQubit(0, 1), ir.GridQubit(1, 0)]


def test_composite_gate_lowering_matches_cirq_matrices():
  """rx/ry/rz, PhasedXPow, FSim, PhasedISwapPow lower to products of power gates."""
  q0, q1 = ir.GridQubit.rect(1, 2)

  def unitary(gates, n):
    flat = ir.Circuit(gates).flat_gates([q0, q1][:n], [])
    u = np.zeros((2**n, 2**n), complex)
    for k, bits in enumerate(itertools.product([0, 1], repeat=n)):
      u[:, k] = O.simulate(n, flat, [], bits).ravel()
    return u

  def same_up_to_phase(a, b):
    i = np.argmax(np.abs(b))
    ph = a.ravel()[i] / b.ravel()[i]
    return np.allclose(a, ph * b, atol=1e-9) and abs(abs(ph) - 1) < 1e-9

  th, ph_, p, t = 0.7, -0.4, 0.3, 0.55
  assert same_up_to_phase(unitary([ir.rx(th)(q0)], 1),
                          np.array([[math.cos(th / 2), -1j * math.sin(th / 2)], [-1j * math.sin(th / 2), math.cos(th / 2)]]))
  zp = np.diag([1, np.exp(1j * math.pi * p)])
  xt = O.gate_matrix(O.GATE_XPOW, t)
  assert same_up_to_phase(unitary(ir.phased_x_pow(q0, p, t), 1), zp @ xt @ zp.conj().T)
  fs = np.array([[1, 0, 0, 0], [0, math.cos(th), -1j * math.sin(th), 0],
                 [0, -1j * math.sin(th), math.cos(th), 0], [0, 0, 0, np.exp(-1j * ph_)]])
  assert same_up_to_phase(unitary(ir.fsim(q0, q1, th, ph_), 2), fs)
  zz = np.kron(zp, zp.conj().T)
  assert same_up_to_phase(unitary(ir.phased_iswap_pow(q0, q1, p, t), 2),
                          zz @ O.gate_matrix(O.GATE_ISWAPPOW, t) @ zz.conj().T)


# ---- energies (tests/models/energy_test.py:113-145,233-249; energy_utils_test.py:86-110) ----
def test_bernoulli_energy():
  e = models.BernoulliEnergy([1, 2, 3])
  with torch.no_grad():
    e.post_process[0].kernel.copy_(torch.tensor([1.0, 1.7, -2.8]))
  bits = torch.tensor([[0, 0, 0], [1, 0, 0], [0, 1, 1]])
  np.testing.assert_allclose(e(bits).detach().numpy(), [-0.1, -2.1, 2.1], atol=1e-6)
  np.testing.assert_allclose(e.logits.detach().numpy(), [2.0, 3.4, -5.6], atol=1e-6)
  qubits = ir.GridQubit.rect(1, 3)
  shards = e.operator_shards(qubits)
  assert [s.masks(qubits) for s in shards] == [[(1.0, 0, 1)], [(1.0, 0, 2)], [(1.0, 0, 4)]]
  assert [s.masks(qubits) for s in shards] == [[tuple(t) for t in op] for op in O.bernoulli_shards(3)]
  assert e.num_bits == 3 and e.bits == [1, 2, 3]
  with pytest.raises(ValueError):
    models.BernoulliEnergy([1, 1])


def test_kobe_energy_and_shards():
  k = models.KOBE([0, 1], 2)
  with torch.no_grad():
    k.post_process[0].kernel.copy_(torch.tensor([1.5, 2.7, -4.0]))
  all_strings = torch.tensor([[0, 0], [0, 1], [1, 0], [1, 1]])
  np.testing.assert_allclose(k(all_strings).detach().numpy(), [0.2, 2.8, 5.2, -8.2], atol=1e-6)
  qubits = ir.GridQubit.rect(1, 4)
  k4 = models.KOBE(list(range(4)), 2)
  assert [s.masks(qubits) for s in k4.operator_shards(qubits)] == [[tuple(t) for t in op] for op in O.kobe_shards(4, 2)]
  np.testing.assert_allclose(k4.operator_expectation(torch.ones(3, 10)).detach().numpy(),
                             np.full(3, k4.post_process[0].kernel.sum().item()), rtol=1e-6)
  with pytest.raises(TypeError):
    models.KOBE([0, 1], 1.5)
  with pytest.raises(ValueError):
    models.KOBE([0, 1], 0)


def test_parity_layer():
  layer = models.Parity([1, 2, 3, 4], 3)
  assert layer.indices == O.parity_indices(4, 3) and layer.num_terms == 14
  out = layer(torch.tensor([[-1, 1, -1, -1]]))
  assert out.tolist() == [[-1, 1, -1, -1] + [-1, 1, 1, -1, -1, 1] + [1, 1, -1, 1]]


def test_hamiltonian_checks_and_shards():
  """hamiltonian.py:41-51."""
  qubits = ir.GridQubit.rect(1, 3)
  circ = models.DirectQuantumCircuit(hea_circuit(qubits, 1, "h"))
  ham = models.Hamiltonian(models.BernoulliEnergy([0, 1, 2]), circ)
  assert len(ham.operator_shards) == 3
  assert ham.circuit_dagger.pqc == circ.pqc**-1
  with pytest.raises(ValueError, match="same number of bits"):
    models.Hamiltonian(models.BernoulliEnergy([0, 1]), circ)


# ---- utils (tests/utils_test.py:47-186) -----------------------------------------------------
def test_unique_expand_weighted_average():
  bits = torch.tensor([[1, 0], [0, 0], [1, 0], [1, 1], [0, 0], [1, 0]], dtype=torch.int8)
  y, idx, counts = utils.unique_bitstrings_with_counts(bits)
  assert y.tolist() == [[1, 0], [0, 0], [1, 1]] and y.dtype == torch.int8
  assert idx.tolist() == [0, 1, 0, 2, 1, 0] and counts.tolist() == [3, 2, 1]
  assert torch.equal(utils.expand_unique_results(y, idx), bits)
  rng = np.random.default_rng(0)
  big = rng.integers(0, 2, size=(500, 9)).astype(np.int8)
  y, idx, counts = utils.unique_bitstrings_with_counts(torch.from_numpy(big))
  wy, widx, wc = O.unique_bitstrings_with_counts(big)
  assert np.array_equal(y.numpy(), wy) and np.array_equal(idx.numpy(), widx) and np.array_equal(counts.numpy(), wc)
  empty = utils.unique_bitstrings_with_counts(torch.zeros((0, 3), dtype=torch.int8))
  assert empty[0].shape == (0, 3) and empty[1].numel() == 0
  np.testing.assert_allclose(
      utils.weighted_average(torch.tensor([1, 3]), torch.tensor([[2.0, 4.0], [6.0, 8.0]])).numpy(), [5.0, 7.0])
  assert utils.Squeeze(1)(torch.zeros(3, 1, 2)).shape == (3, 2)


def test_device_side_unique_equals_the_host_path():
  """`utils._unique_on_device` (torch ops on the tensor's own device: what CUDA bitstrings go through) returns the same
  bits as the numpy path on the reference's own cases (tests/utils_test.py:107-186: `test_short`, `test_long`) and on
  random, ragged and non-binary inputs; a tensor the function just returned is recognised and not sorted again."""
  short = [[1], [0], [0], [1], [1], [0], [1], [1]]
  long_ = [[1, 0, 1], [1, 1, 1], [0, 1, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1], [1, 0, 1], [1, 0, 1]]
  for rows, want_y, want_idx, want_c in ((short, [[1], [0]], [0, 1, 1, 0, 0, 1, 0, 0], [5, 3]),
                                         (long_, [[1, 0, 1], [1, 1, 1], [0, 1, 1]], [0, 1, 2, 0, 1, 2, 0, 0], [4, 2, 2])):
    for fn in (utils.unique_bitstrings_with_counts, lambda t: utils._unique_on_device(t, torch.int32)):
      y, idx, c = fn(torch.tensor(rows, dtype=torch.int8))
      assert y.tolist() == want_y and idx.tolist() == want_idx and c.tolist() == want_c
      assert utils.expand_unique_results(y, idx).tolist() == rows
  rng = np.random.default_rng(3)
  for count, n, hi in [(1, 1, 2), (4096, 20, 2), (1024, 12, 2), (700, 5, 2), (300, 4, 5), (100, 70, 2), (64, 62, 2), (9, 63, 2)]:
    arr = rng.integers(0, hi, size=(count, n)).astype(np.int8)
    t = torch.from_numpy(arr)
    host = utils.unique_bitstrings_with_counts(t)
    dev = utils._unique_on_device(t, torch.int32)
    want = O.unique_bitstrings_with_counts(arr)
    for a, b, w in zip(host, dev, want):
      assert torch.equal(a, b) and np.array_equal(a.numpy(), w) and a.dtype == b.dtype, (count, n, hi)
  y, idx, c = utils.unique_bitstrings_with_counts(torch.from_numpy(rng.integers(0, 2, size=(50, 4)).astype(np.int8)))
  again = utils.unique_bitstrings_with_counts(y)
  assert again[0] is y and again[1].tolist() == list(range(y.shape[0])) and bool((again[2] == 1).all())
  y.add_(0)                                            # any in-place write voids the mark
  assert utils.unique_bitstrings_with_counts(y)[0] is not y
  assert utils.unique_bitstrings_with_counts(y.clone())[0].shape == y.shape


def test_analytic_inference_needs_gpu_and_says_so():
  """The product fails loudly without the HIP device -- no CPU fallback."""
  if torch.cuda.is_available():
    pytest.skip("GPU present")
  from qhbmlib_amd import EngineError, inference
  qubits = ir.GridQubit.rect(1, 2)
  circ = models.DirectQuantumCircuit(hea_circuit(qubits, 1, "g"))
  qnn = inference.AnalyticQuantumInference(circ)
  with pytest.raises(EngineError, match="no CPU fallback"):
    qnn.expectation(torch.zeros((2, 2), dtype=torch.int8), [ir.PZ(qubits[0]) + ir.PZ(qubits[1])])
