"""Reference-style tests of the host API on the MI355X engine (GPU only).

These read like the reference's own tests (file:line cited); expected values are
its closed forms or the numpy oracle.  Tolerances: analytic atol 2e-3 is the
reference's bar (tests/inference/qnn_test.py:49) -- the engine is held to 1e-5
against the oracle; sample-limited losses use the reference's rtol 3e-2.
"""
import itertools
import math

import numpy as np
import pytest
import torch

from oracle import qhbm_oracle as O
from qhbmlib_amd import data, inference, ir, models
from tests.test_host_api import hea_circuit

pytestmark = pytest.mark.gpu


def _set(param, values):
  with torch.no_grad():
    param.copy_(torch.as_tensor(np.asarray(values), dtype=torch.float32))


def _jacobian(fn, variables):
  out = fn()
  flat = out.reshape(-1)
  rows = []
  for i in range(flat.numel()):
    grads = torch.autograd.grad(flat[i], variables, retain_graph=True, allow_unused=True)
    rows.append([torch.zeros_like(v) if g is None else g for g, v in zip(grads, variables)])
  return out, [torch.stack([r[k] for r in rows]).reshape(out.shape + variables[k].shape).cpu().numpy()
               for k in range(len(variables))]


# ---- tests/inference/qnn_test.py:83-180 ------------------------------------------------------
def test_expectation_x_pow():
  num_bits = 3
  qubits = ir.GridQubit.rect(1, num_bits)
  p_circuit = ir.Circuit(ir.X(q)**ir.Symbol("p") for q in qubits)
  p_qnn = models.DirectQuantumCircuit(p_circuit, name="p_qnn")
  _set(p_qnn.trainable_variables[0], [0.37])
  initial_states = torch.tensor(5 * list(itertools.product([0, 1], repeat=num_bits)), dtype=torch.int8)
  sin_pi_p, cos_pi_p = math.sin(math.pi * 0.37), math.cos(math.pi * 0.37)
  qnn = inference.AnalyticQuantumInference(p_qnn)
  for pauli, val, grad in ((ir.PX, lambda s: 0.0, lambda s: 0.0),
                           (ir.PY, lambda s: -((-1.0)**s) * sin_pi_p, lambda s: -((-1.0)**s) * math.pi * cos_pi_p),
                           (ir.PZ, lambda s: ((-1.0)**s) * cos_pi_p, lambda s: -((-1.0)**s) * math.pi * sin_pi_p)):
    ops = [1.0 * pauli(q) for q in qubits]
    actual, (jac,) = _jacobian(lambda: qnn.expectation(initial_states, ops), p_qnn.trainable_variables)
    expected = [[val(s) for s in bits] for bits in initial_states.tolist()]
    expected_grad = [[grad(s) for s in bits] for bits in initial_states.tolist()]
    assert actual.shape == (40, 3)
    np.testing.assert_allclose(actual.detach().cpu().numpy(), expected, atol=1e-5)
    np.testing.assert_allclose(jac[:, :, 0], expected_grad, atol=2e-5)


# ---- tests/inference/qnn_test.py:183-264 (oracle instead of cirq.Simulator) -------------------
def test_expectation_general_circuit_vs_oracle():
  n = 3
  qubits = ir.GridQubit.rect(1, n)
  a, b, c, d = ir.symbols("aa bb cc dd")
  raw = ir.Circuit(ir.H(qubits[0])**a, ir.CNOT(qubits[0], qubits[1])**b, ir.YY(qubits[1], qubits[2])**c,
                   ir.ISWAP(qubits[0], qubits[2])**d, ir.rx(a)(qubits[1]), ir.fsim(qubits[1], qubits[2], 0.3, c),
                   ir.SWAP(qubits[0], qubits[1])**0.37, ir.ZZ(qubits[0], qubits[2])**b)
  circ = models.DirectQuantumCircuit(raw)
  values = np.array([0.31, -0.62, 0.45, 0.9])
  _set(circ.trainable_variables[0], values)
  ops = [ir.PauliSum.from_pauli_strings([ir.PZ(q) for q in qubits]),
         ir.PX(qubits[0]) * ir.PY(qubits[2]) + 0.5 * ir.PZ(qubits[1])]
  all_bits = torch.tensor(list(itertools.product([0, 1], repeat=n)), dtype=torch.int8)
  qnn = inference.AnalyticQuantumInference(circ)
  actual, (jac,) = _jacobian(lambda: qnn.expectation(all_bits, ops), circ.trainable_variables)
  flat = raw.flat_gates(circ.qubits, circ.symbol_names)
  want, want_jac = O.expectation_jacobian(n, flat, values, all_bits.numpy(), [op.masks(qubits) for op in ops])
  np.testing.assert_allclose(actual.detach().cpu().numpy(), want, atol=1e-5)
  np.testing.assert_allclose(jac, want_jac, atol=1e-4)
  # parameter-shift gradients are not available for ISWAP: loud error, adjoint is the default
  shift = inference.AnalyticQuantumInference(circ, gradient_method=1)
  with pytest.raises(Exception, match="ISWAPPOW"):
    shift.expectation(all_bits, ops).sum().backward()


# ---- tests/inference/qnn_test.py:266-369: modular Hamiltonian branch -----------------------------
@pytest.mark.parametrize("energy_class,energy_args", [(models.BernoulliEnergy, []), (models.KOBE, [2])])
def test_expectation_modular_hamiltonian(energy_class, energy_args):
  n = 3
  qubits = ir.GridQubit.rect(1, n)
  energy_h = energy_class(*([list(range(n))] + energy_args))
  circuit_h = models.DirectQuantumCircuit(hea_circuit(qubits, 2, "h"))
  rng = np.random.default_rng(11)
  h_vals = rng.uniform(-1, 1, len(circuit_h.symbol_names))
  thetas = rng.uniform(-1, 1, energy_h.post_process[0].kernel.numel())
  _set(circuit_h.trainable_variables[0], h_vals)
  _set(energy_h.post_process[0].kernel, thetas)
  hamiltonian_measure = models.Hamiltonian(energy_h, circuit_h)
  model_circuit = models.DirectQuantumCircuit(hea_circuit(qubits, 2, "m"))
  m_vals = rng.uniform(-1, 1, len(model_circuit.symbol_names))
  _set(model_circuit.trainable_variables[0], m_vals)
  all_bits = torch.tensor(list(itertools.product([0, 1], repeat=n)), dtype=torch.int8)
  qnn = inference.AnalyticQuantumInference(model_circuit)
  variables = energy_h.trainable_variables + circuit_h.trainable_variables + model_circuit.trainable_variables
  actual, jacs = _jacobian(lambda: qnn.expectation(all_bits, hamiltonian_measure), variables)
  assert actual.shape == (8, 1)
  # oracle: params = [model..., hamiltonian...]
  m_gates = model_circuit.pqc.flat_gates(qubits, model_circuit.symbol_names + circuit_h.symbol_names)
  h_gates = circuit_h.pqc.flat_gates(qubits, model_circuit.symbol_names + circuit_h.symbol_names)
  params = np.concatenate([m_vals, h_vals])
  shards = O.bernoulli_shards(n) if energy_class is models.BernoulliEnergy else O.kobe_shards(n, 2)
  total = m_gates + O.inverse_gates(h_gates)
  shard_vals, shard_jac = O.expectation_jacobian(n, total, params, all_bits.numpy(), shards)
  np.testing.assert_allclose(actual.detach().cpu().numpy()[:, 0], shard_vals @ thetas, atol=2e-5)
  np.testing.assert_allclose(jacs[0][:, 0, :], shard_vals, atol=2e-5)                      # d/d theta
  want_phi = np.einsum("t,btp->bp", thetas, shard_jac)
  np.testing.assert_allclose(jacs[1][:, 0, :], want_phi[:, len(m_vals):], atol=2e-4)        # d/d phi (H)
  np.testing.assert_allclose(jacs[2][:, 0, :], want_phi[:, :len(m_vals)], atol=2e-4)        # d/d phi (model)


def test_type_error_for_non_pauli_hamiltonian():
  """qnn.py:128-130."""
  qubits = ir.GridQubit.rect(1, 2)
  circ = models.DirectQuantumCircuit(hea_circuit(qubits, 1, "t"))
  general = models.BitstringEnergy([0, 1], [models.SpinsFromBitstrings(), models.VariableDot()])
  ham = models.Hamiltonian(general, models.DirectQuantumCircuit(hea_circuit(qubits, 1, "other")))
  with pytest.raises(TypeError, match="General Hamiltonians not accepted"):
    inference.AnalyticQuantumInference(circ).expectation(torch.zeros((1, 2), dtype=torch.int8), ham)


def test_tfq_compat_bit_order_n12():
  """both bit-order modes against the oracle at n = 12."""
  n = 12
  qubits = ir.GridQubit.rect(1, n)
  rng = np.random.default_rng(2)
  bits = rng.integers(0, 2, size=(5, n)).astype(np.int8)
  op = ir.PauliSum.from_pauli_strings([ir.PZ(q) * (i + 1.0) for i, q in enumerate(qubits)])
  for compat in (False, True):
    circ = models.DirectQuantumCircuit(hea_circuit(qubits, 1, "q"), tfq_compat_bit_order=compat)
    vals = rng.uniform(-1, 1, len(circ.symbol_names))
    _set(circ.trainable_variables[0], vals)
    got = inference.AnalyticQuantumInference(circ).expectation(torch.from_numpy(bits), [op])
    flat = circ.pqc.flat_gates(qubits, circ.symbol_names)
    want = O.expectation(n, flat, vals, bits, [op.masks(qubits)], tfq_compat_bit_order=compat)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want, atol=1e-4)


# ---- tests/inference/vqt_loss_test.py:133-205 ---------------------------------------------------
@pytest.mark.parametrize("num_qubits", [1, 2])
def test_vqt_loss_value_x_rot(num_qubits):
  num_samples = int(2e5)
  close_rtol = 3e-2
  rng = np.random.default_rng(5 + num_qubits)
  energy = models.BernoulliEnergy(list(range(num_qubits)))
  thetas = rng.uniform(0.5, 2.0, num_qubits) * rng.choice([-1, 1], num_qubits)
  _set(energy.post_process[0].kernel, thetas)
  e_infer = inference.BernoulliEnergyInference(energy, num_samples, initial_seed=7)
  qubits = ir.GridQubit.rect(1, num_qubits)
  r_circuit = ir.Circuit(ir.rx(ir.Symbol(f"phi_{n}"))(q) for n, q in enumerate(qubits))
  circuit = models.DirectQuantumCircuit(r_circuit)
  phis = rng.uniform(0.3, 1.0, num_qubits) * rng.choice([-1, 1], num_qubits)
  _set(circuit.trainable_variables[0], phis)
  q_infer = inference.AnalyticQuantumInference(circuit)
  qhbm_infer = inference.QHBM(e_infer, q_infer)
  test_h = [ir.PauliSum.from_pauli_strings([ir.PY(q) for q in qubits])]
  beta = 1.7
  expected_expectation = np.sum(np.tanh(thetas) * np.sin(phis))
  np.testing.assert_allclose(qhbm_infer.expectation(test_h)[0].item(), expected_expectation, rtol=close_rtol)
  expected_entropy = np.sum(-thetas * np.tanh(thetas) + np.log(2 * np.cosh(thetas)))
  np.testing.assert_allclose(e_infer.entropy().item(), expected_entropy, rtol=1e-5)
  loss = inference.vqt(qhbm_infer, test_h, beta)
  np.testing.assert_allclose(loss.item(), beta * expected_expectation - expected_entropy, rtol=close_rtol)
  test_thetas, test_phis = energy.trainable_variables[0], circuit.trainable_variables[0]
  g_thetas, g_phis = torch.autograd.grad(loss, (test_thetas, test_phis))
  np.testing.assert_allclose(g_thetas.cpu().numpy(), (1 - np.tanh(thetas)**2) * (beta * np.sin(phis) + thetas),
                             rtol=close_rtol, atol=5e-3)
  np.testing.assert_allclose(g_phis.cpu().numpy(), beta * np.tanh(thetas) * np.cos(phis), rtol=close_rtol, atol=5e-3)


# ---- tests/inference/vqt_loss_test.py:46-83 -----------------------------------------------------
def test_self_vqt():
  """VQT of a model against a Hamiltonian with the same weights: loss = -log Z, grads ~ 0."""
  n = 2
  qubits = ir.GridQubit.rect(1, n)
  rng = np.random.default_rng(0)
  theta_vals = rng.uniform(-1, 1, 3)

  def make(name):
    energy = models.KOBE(list(range(n)), n)
    _set(energy.post_process[0].kernel, theta_vals)
    circuit = models.DirectQuantumCircuit(hea_circuit(qubits, 3, name))
    return energy, circuit

  data_energy, data_circuit = make("data")
  model_energy, model_circuit = make("model")
  phi_vals = rng.uniform(-1, 1, len(model_circuit.symbol_names))
  _set(data_circuit.trainable_variables[0], phi_vals)
  _set(model_circuit.trainable_variables[0], phi_vals)
  data_h = models.Hamiltonian(data_energy, data_circuit)
  e_infer = inference.AnalyticEnergyInference(model_energy, int(2e5), initial_seed=3)
  model_infer = inference.QHBM(e_infer, inference.AnalyticQuantumInference(model_circuit))
  loss = inference.vqt(model_infer, data_h, 1.0)
  np.testing.assert_allclose(loss.item(), -e_infer.log_partition().item(), atol=2e-3)
  grads = torch.autograd.grad(loss, model_energy.trainable_variables + model_circuit.trainable_variables)
  for g in grads:
    np.testing.assert_allclose(g.cpu().numpy(), 0.0, atol=2e-2)


# ---- tests/inference/qmhl_loss_test.py:136-272 --------------------------------------------------
@pytest.mark.parametrize("num_qubits", [1, 2])
def test_qmhl_loss_value_x_rot(num_qubits):
  num_samples = int(2e5)
  close_rtol = 3e-2
  rng = np.random.default_rng(20 + num_qubits)
  energy = models.BernoulliEnergy(list(range(num_qubits)))
  thetas = rng.uniform(0.25, 1.0, num_qubits)
  _set(energy.post_process[0].kernel, thetas)
  e_infer = inference.BernoulliEnergyInference(energy, num_samples, initial_seed=5)
  qubits = ir.GridQubit.rect(1, num_qubits)
  circuit = models.DirectQuantumCircuit(ir.Circuit(ir.rx(ir.Symbol(f"phi_{n}"))(q) for n, q in enumerate(qubits)))
  phis = rng.uniform(math.pi / 4, math.pi, num_qubits)
  _set(circuit.trainable_variables[0], phis)
  qhbm_infer = inference.QHBM(e_infer, inference.AnalyticQuantumInference(circuit))
  alphas = rng.uniform(-math.pi, math.pi, num_qubits)
  data_circuit = models.DirectQuantumCircuit(ir.Circuit(ir.ry(float(a))(q) for a, q in zip(alphas, qubits)))
  data_q_infer = inference.AnalyticQuantumInference(data_circuit)
  data_probs = rng.uniform(0, 1, num_qubits)
  gen = torch.Generator().manual_seed(9)
  data_samples = torch.bernoulli(torch.tensor(1 - data_probs).expand(num_samples, -1), generator=gen).to(torch.int8)

  class FixedData(data.QuantumData):
    def __init__(self, samples, q_infer):
      self.samples, self.q_infer = samples, q_infer

    def expectation(self, observable):
      return torch.mean(self.q_infer.expectation(self.samples, observable))

  actual_data = FixedData(data_samples, data_q_infer)
  loss = inference.qmhl(actual_data, qhbm_infer)
  expected_expectation = np.sum(thetas * (2 * data_probs - 1) * np.cos(alphas) * np.cos(phis))
  expected_log_partition = np.sum(np.log(2 * np.cosh(thetas)))
  np.testing.assert_allclose(loss.item(), expected_expectation + expected_log_partition, rtol=close_rtol, atol=5e-3)
  g_thetas, g_phis = torch.autograd.grad(loss, (energy.trainable_variables[0], circuit.trainable_variables[0]))
  np.testing.assert_allclose(g_thetas.cpu().numpy(),
                             (2 * data_probs - 1) * np.cos(alphas) * np.cos(phis) + np.tanh(thetas),
                             rtol=close_rtol, atol=5e-3)
  np.testing.assert_allclose(g_phis.cpu().numpy(), -thetas * (2 * data_probs - 1) * np.cos(alphas) * np.sin(phis),
                             rtol=close_rtol, atol=5e-3)


# ---- QAIA ansatz on the engine (circuit.py:211-292, used at baselines/train.py:139-143) ----
def test_qaia_expectation_and_tied_gradients_vs_oracle():
  n, num_layers = 4, 2
  qubits = ir.GridQubit.rect(1, n)
  classical = [ir.PZ(q) for q in qubits] + [ir.PZ(a) * ir.PZ(b) for a, b in zip(qubits, qubits[1:])]
  x_terms, zz_xx = ir.PauliSum(), ir.PauliSum()
  for q in qubits:
    x_terms += ir.PX(q)
  for a, b in zip(qubits[::2], qubits[1::2]):
    zz_xx += ir.PY(a) * ir.PY(b)
  quantum = [x_terms, zz_xx]
  qaia = models.QAIA(quantum, classical, num_layers)
  rng = np.random.default_rng(17)
  etas, thetas, gammas = qaia.value_layers_inputs[0]
  _set(etas, rng.uniform(-1, 1, etas.shape))
  _set(thetas, rng.uniform(-1, 1, thetas.shape))
  _set(gammas, rng.uniform(-1, 1, gammas.shape))
  xxz = ir.PauliSum()
  for a, b in zip(qubits, qubits[1:]):
    xxz += ir.PX(a) * ir.PX(b) + ir.PY(a) * ir.PY(b) + 0.5 * ir.PZ(a) * ir.PZ(b)
  ops = [xxz, ir.as_pauli_sum(1.0 * ir.PZ(qubits[1]) * ir.PX(qubits[3]))]
  all_bits = torch.tensor(list(itertools.product([0, 1], repeat=n)), dtype=torch.int8)
  qnn = inference.AnalyticQuantumInference(qaia)
  actual, jacs = _jacobian(lambda: qnn.expectation(all_bits, ops), qaia.trainable_variables)
  values = qaia.symbol_values.detach().cpu().numpy().astype(np.float64)
  flat = qaia.pqc.flat_gates(qaia.qubits, qaia.symbol_names)
  want, want_jac = O.expectation_jacobian(n, flat, values, all_bits.numpy(), [op.masks(qubits) for op in ops])
  np.testing.assert_allclose(actual.detach().cpu().numpy(), want, atol=2e-5)
  # chain rule through embed_params: values per layer are [eta_l * theta_b ..., gamma_l_r ...]
  c, q = len(classical), len(quantum)
  per_layer = want_jac.reshape(want.shape + (num_layers, c + q))
  e, t = etas.detach().numpy().astype(np.float64), thetas.detach().numpy().astype(np.float64)
  want_eta = np.einsum("btlc,c->btl", per_layer[..., :c], t)
  want_theta = np.einsum("btlc,l->btc", per_layer[..., :c], e)
  want_gamma = per_layer[..., c:]
  np.testing.assert_allclose(jacs[0], want_eta, atol=2e-4)
  np.testing.assert_allclose(jacs[1], want_theta, atol=2e-4)
  np.testing.assert_allclose(jacs[2], want_gamma, atol=2e-4)


def test_more_observables_than_one_engine_call_holds():
  """A third-order KOBE on 14 qubits has 469 shards; with the per-call limit lowered to 200 the
  host slices the list over three engine calls -- values and gradients must not notice."""
  n = 14
  qubits = ir.GridQubit.rect(1, n)
  energy_h = models.KOBE(list(range(n)), 3)
  rng = np.random.default_rng(4)
  thetas = rng.uniform(-1, 1, energy_h.post_process[0].kernel.numel())
  _set(energy_h.post_process[0].kernel, thetas)
  circuit_h = models.DirectQuantumCircuit(hea_circuit(qubits, 1, "h"))
  model_circuit = models.DirectQuantumCircuit(hea_circuit(qubits, 1, "m"))
  _set(circuit_h.trainable_variables[0], rng.uniform(-1, 1, len(circuit_h.symbol_names)))
  _set(model_circuit.trainable_variables[0], rng.uniform(-1, 1, len(model_circuit.symbol_names)))
  ham = models.Hamiltonian(energy_h, circuit_h)
  states = torch.from_numpy(rng.integers(0, 2, size=(3, n)).astype(np.int8))
  results = []
  for limit in (1024, 200):
    qnn = inference.AnalyticQuantumInference(model_circuit)
    qnn.MAX_OPS_PER_CALL = limit
    for v in (energy_h.trainable_variables + circuit_h.trainable_variables + model_circuit.trainable_variables):
      v.grad = None
    out = qnn.expectation(states, ham)
    out.sum().backward()
    results.append((out.detach().cpu().numpy(), energy_h.post_process[0].kernel.grad.cpu().numpy().copy(),
                    model_circuit.trainable_variables[0].grad.cpu().numpy().copy(), len(qnn._engines)))
  assert results[0][3] == 1 and results[1][3] == 3
  np.testing.assert_allclose(results[1][0], results[0][0], atol=1e-5)
  np.testing.assert_allclose(results[1][1], results[0][1], atol=1e-5)
  np.testing.assert_allclose(results[1][2], results[0][2], atol=1e-4)


def test_autograd_backward_twice_and_interleaved_engines():
  """The autograd function starts the backward sweep from the states its forward left in the
  engine; a second backward (retain_graph) or an interleaved call must transparently re-simulate."""
  n = 5
  qubits = ir.GridQubit.rect(1, n)
  circ = models.DirectQuantumCircuit(hea_circuit(qubits, 2, "a"))
  _set(circ.trainable_variables[0], np.random.default_rng(2).uniform(-1, 1, len(circ.symbol_names)))
  ops = [ir.PauliSum.from_pauli_strings([ir.PZ(q) for q in qubits])]
  states = torch.tensor(list(itertools.product([0, 1], repeat=n))[:7], dtype=torch.int8)
  qnn = inference.AnalyticQuantumInference(circ)
  out = qnn.expectation(states, ops)
  (g1,) = torch.autograd.grad(out.sum(), circ.trainable_variables, retain_graph=True)   # retained states
  (g2,) = torch.autograd.grad(out.sum(), circ.trainable_variables, retain_graph=True)   # consumed: full VJP
  np.testing.assert_allclose(g1.cpu().numpy(), g2.cpu().numpy(), atol=1e-6)
  out_a = qnn.expectation(states, ops)
  out_b = qnn.expectation(states[:3], ops)          # same engine: drops out_a's states
  (ga,) = torch.autograd.grad(out_a.sum(), circ.trainable_variables)
  (gb,) = torch.autograd.grad(out_b.sum(), circ.trainable_variables)
  np.testing.assert_allclose(ga.cpu().numpy(), g1.cpu().numpy(), atol=1e-6)
  flat = circ.pqc.flat_gates(circ.qubits, circ.symbol_names)
  _, jac = O.expectation_jacobian(n, flat, circ.symbol_values.detach().numpy().astype(np.float64),
                                  states[:3].numpy(), [op.masks(qubits) for op in ops])
  np.testing.assert_allclose(gb.cpu().numpy(), jac.sum((0, 1)), atol=1e-4)


# ---- engine cache keyed by content (id()-keyed cache returned stale engines) ---
def test_fresh_operator_lists_never_hit_a_stale_engine():
  """Rebuilds X, Y, Z operator lists (and Hamiltonian objects) every iteration on ONE qnn --
  CPython hands the freed lists' ids to the new ones -- and checks the closed forms of
  tests/inference/qnn_test.py:83-180 every time."""
  num_bits = 3
  qubits = ir.GridQubit.rect(1, num_bits)
  p_qnn = models.DirectQuantumCircuit(ir.Circuit(ir.X(q)**ir.Symbol("p") for q in qubits), name="p_qnn")
  _set(p_qnn.trainable_variables[0], [0.37])
  states = torch.tensor(list(itertools.product([0, 1], repeat=num_bits)), dtype=torch.int8)
  sin_pi_p, cos_pi_p = math.sin(math.pi * 0.37), math.cos(math.pi * 0.37)
  qnn = inference.AnalyticQuantumInference(p_qnn)
  for _ in range(10):
    for pauli, val in ((ir.PX, lambda s: 0.0), (ir.PY, lambda s: -((-1.0)**s) * sin_pi_p),
                       (ir.PZ, lambda s: ((-1.0)**s) * cos_pi_p)):
      got = qnn.expectation(states, [1.0 * pauli(q) for q in qubits]).detach().cpu().numpy()
      np.testing.assert_allclose(got, [[val(s) for s in bits] for bits in states.tolist()], atol=1e-5)
    # in-place mutation of a PauliSum between calls
    op = ir.PauliSum.from_pauli_strings([ir.PZ(qubits[0])])
    z0 = qnn.expectation(states, [op]).detach().cpu().numpy()[:, 0]
    op += ir.PZ(qubits[1])
    z01 = qnn.expectation(states, [op]).detach().cpu().numpy()[:, 0]
    np.testing.assert_allclose(z0, [((-1.0)**b[0]) * cos_pi_p for b in states.tolist()], atol=1e-5)
    np.testing.assert_allclose(z01, [(((-1.0)**b[0]) + ((-1.0)**b[1])) * cos_pi_p for b in states.tolist()], atol=1e-5)
    # fresh Hamiltonian objects with different thetas: identity circuit, Bernoulli energy
    for theta in ([1.0, 0.0, 0.0], [0.0, 2.0, 0.0], [0.0, 0.0, -3.0]):
      e = models.BernoulliEnergy(list(range(num_bits)))
      _set(e.post_process[0].kernel, theta)
      ident = models.DirectQuantumCircuit(ir.Circuit(ir.X(q)**(0.0 * ir.Symbol(f"i{q.col}")) for q in qubits))
      ham = models.Hamiltonian(e, ident)
      got = qnn.expectation(states, ham).detach().cpu().numpy()[:, 0]
      k = int(np.argmax(np.abs(theta)))
      np.testing.assert_allclose(got, [theta[k] * ((-1.0)**b[k]) * cos_pi_p for b in states.tolist()], atol=1e-5)
  assert len(qnn._engines) <= qnn._engines.max_engines  # LRU bound


def test_autograd_through_a_batch_larger_than_one_backward_chunk():
  """with more unique states than one backward chunk holds the C side retains nothing;
  the autograd function must notice and run the full VJP instead of raising."""
  n = 5
  qubits = ir.GridQubit.rect(1, n)
  circ = models.DirectQuantumCircuit(hea_circuit(qubits, 2, "c"))
  vals = np.random.default_rng(3).uniform(-1, 1, len(circ.symbol_names))
  _set(circ.trainable_variables[0], vals)
  ops = [ir.PauliSum.from_pauli_strings([ir.PZ(q) for q in qubits]), ir.PX(qubits[0]) * ir.PX(qubits[1])]
  states = torch.tensor(list(itertools.product([0, 1], repeat=n)), dtype=torch.int8)
  qnn = inference.AnalyticQuantumInference(circ)
  qnn.expectation(states[:1], ops)  # creates the engine
  (eng,) = list(qnn._engines._engines.values())
  eng.set_option("chunk_states", 5)
  out = qnn.expectation(states, ops)
  assert eng.retained is None and eng.retained_states() == 0
  w = torch.arange(out.numel(), dtype=torch.float32, device=out.device).reshape(out.shape) / out.numel()
  (g,) = torch.autograd.grad((out * w).sum(), circ.trainable_variables)
  flat = circ.pqc.flat_gates(circ.qubits, circ.symbol_names)
  want, jac = O.expectation_jacobian(n, flat, vals, states.numpy(), [ir.as_pauli_sum(op).masks(qubits) for op in ops])
  np.testing.assert_allclose(out.detach().cpu().numpy(), want, atol=1e-5)
  np.testing.assert_allclose(g.cpu().numpy(), np.einsum("bt,btp->p", w.cpu().numpy(), jac), atol=1e-4)
  eng.set_option("chunk_states", 0)
  out = qnn.expectation(states, ops)
  assert eng.retained is not None and eng.retained_states() == states.shape[0]


def test_qmhl_with_a_fixed_data_qhbm_skips_its_gradient_work():
  """QMHL against data from a FIXED QHBM (its circuit's variables do not require grad): the total circuit of
  `data.expectation(model.modular_hamiltonian)` is U_data then U_model^dagger, and only the second half's symbols
  want a gradient.  The host derives the engine's gradient mask from requires_grad (`symbol_requires_grad`); the
  model's gradients must equal those of the unmasked run, the data circuit gets none, and the backward plan is
  not more expensive than the unmasked one."""
  qdata = data
  n, layers = 13, 3
  qubits = ir.GridQubit.rect(1, n)

  def make(tag, seed):
    torch.manual_seed(seed)
    ebm = models.KOBE(list(range(n)), 2)
    ebm.build([None, n])
    with torch.no_grad():
      ebm.trainable_variables[0].uniform_(-0.3, 0.3)
    circ = models.DirectQuantumCircuit(hea_circuit(qubits, layers, tag))
    with torch.no_grad():
      circ.trainable_variables[0].uniform_(-1.0, 1.0)
    e_inf = inference.AnalyticEnergyInference(ebm, 64, initial_seed=seed)
    return inference.QHBM(e_inf, inference.AnalyticQuantumInference(circ)), ebm, circ

  target, _, data_circ = make("fd", 3)
  model, ebm, circ = make("fm", 5)
  data_params = list(data_circ.trainable_variables)
  results = []
  for freeze in (False, True):
    for p in data_params:
      p.requires_grad_(not freeze)
      p.grad = None
    for p in list(ebm.parameters()) + circ.trainable_variables:
      p.grad = None
    loss = inference.qmhl(qdata.QHBMData(target), model)
    loss.backward()
    (eng,) = list(target.q_inference._engines._engines.values())
    results.append((float(loss.detach()), circ.trainable_variables[0].grad.clone(),
                    [p.grad.clone() for p in ebm.parameters() if p.grad is not None],
                    eng.flop_model(1, True)["bwd_flops"], data_params[0].grad))
  (l0, g0, e0, f0, d0), (l1, g1, e1, f1, d1) = results
  assert d0 is not None and d1 is None
  assert f1 <= f0
  assert l0 == l1   # (fixed seeds: the same samples both times)
  np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), atol=2e-5 * max(1.0, float(g0.abs().max())))
  for a, b in zip(e0, e1):
    np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), atol=2e-5 * max(1.0, float(a.abs().max())))
