"""The HIP engine against oracle fixtures AT the BASELINE.json sizes (GPU only).

tests/golden/make_golden_large.py generated these in the build container with the numpy
complex128 oracle and the independent C fp32 restatement (each cross-checked against the other
there); the multi-pass plans, zero-tile skipping and chunking that only exist at n >= 15 are
compared here with an INDEPENDENT simulator, not with the engine itself (the pattern of
/root/reference/tests/inference/qnn_test.py:183-264, which compares against cirq.Simulator).

Tolerances: values 5e-5 * sum|c_k| at n = 20 depth 16 (measured error is ~20x
smaller); gradients 1e-4 * max(1, |grad|_inf); single Pauli strings 5e-5.
"""
import numpy as np
import pytest
import torch

from qhbmlib_amd import _engine as E
from tests import golden_util as G

pytestmark = pytest.mark.gpu


def _engine(n, gates, n_params, ops, **opts):
  eng = E.Engine(0)
  for k, v in opts.items():
    eng.set_option(k, v)
  eng.set_circuit(n, gates, n_params)
  eng.set_observables(ops)
  return eng


@pytest.fixture(scope="module")
def c3():
  return G.load("c3_n20_l16.npz")


@pytest.mark.parametrize("opts", [{}, {"chunk_states": 3}, {"tile_qubits": 12, "adjoint_tile_qubits": 11},
                                  {"tile_qubits": 14, "adjoint_tile_qubits": 13}, {"cph_wave_bits": 0},
                                  {"values_from_observable": 0}],
                         ids=["default-plan", "chunked", "small-tiles", "large-tiles", "unpruned-layout",
                              "measured-values"])
def test_c3_values_and_vjp_against_oracle(c3, opts):
  """BASELINE config 3's circuit (20 qubits, depth 16, 944 parameters, XXZ): default 9 / 11-pass
  plans, a chunked batch and two other tile geometries, on |0..0>, |1..1> and two random states; the
  plain layout without dead waves, tail tiles and the searched pass order ("unpruned-layout": the
  pruned default must not differ from it beyond the tolerance -- a single observable, so the values
  come from lambda = O psi and the rows are scaled afterwards) and values measured in the forward sweep."""
  n, gates, ops = int(c3["n"]), G.gates_of(c3["gates"]), G.ops_of(c3["ops"])
  norm = sum(abs(c) for c, _, _ in ops[0])
  eng = _engine(n, gates, len(c3["params"]), ops, **opts)
  bits, params = c3["bits"], c3["params"]
  vals = eng.expectation(bits, params).cpu().numpy()[:, 0]
  assert np.abs(vals - c3["values"]).max() <= 5e-5 * norm, np.abs(vals - c3["values"]).max()
  up = np.array([[0.7], [-0.4], [0.25], [1.1]], np.float32)
  want = (up * c3["grads"]).sum(0)
  tol = 1e-4 * max(1.0, np.abs(want).max())
  vals2, grad = eng.expectation_vjp(bits, params, up)
  assert np.abs(vals2.cpu().numpy()[:, 0] - c3["values"]).max() <= 5e-5 * norm
  assert np.abs(grad.cpu().numpy() - want).max() <= tol, np.abs(grad.cpu().numpy() - want).max()
  # forward now, backward later from the retained states (what the autograd function does)
  eng.expectation(bits, params, retain=True)
  if eng.retained is not None:
    grad_r = eng.expectation_vjp_retained(bits, params, up)
    assert np.abs(grad_r.cpu().numpy() - want).max() <= tol
  else:
    assert opts.get("chunk_states", 0) and opts["chunk_states"] < len(bits)


@pytest.mark.parametrize("opts", [{}, {"gather_multi_values": 1}, {"multi_observable_values": 0}, {"observable_kernel": 0}],
                         ids=["values-from-the-block-kernel", "values-and-lambda-from-one-gather-launch", "measured-in-the-passes",
                              "gather-kernel"])
def test_c3_three_observables_values_and_vjp_against_oracle(c3, opts):
  """Config 3's circuit with the XXZ chain as THREE observables (XX, YY, ZZ sums) -- several operators per call, the
  reference's normal usage (tests/inference/qnn_test.py:187-190,266-369): lean forward passes + one launch for the
  three values, one for lambda = sum_t upstream_t O_t psi.  Against the committed fixture (the three values add up to
  the fixture's XXZ value; with equal upstream weights the VJP is the fixture's) and, per observable with unequal
  weights, against the C oracle run live on the host."""
  from oracle import qhbm_cpu as C
  n, gates, ops = int(c3["n"]), G.gates_of(c3["gates"]), G.ops_of(c3["ops"])
  whole = ops[0]
  ops3 = [whole[0::3], whole[1::3], whole[2::3]]
  assert all(x != 0 and z == 0 for _, x, z in ops3[0]) and all(x == z != 0 for _, x, z in ops3[1]) and all(x == 0 for _, x, z in ops3[2])
  eng = _engine(n, gates, len(c3["params"]), ops3, **opts)
  if opts.get("gather_multi_values"):  # round 5: the gather kernel with a value accumulator per observable (on request)
    assert "values = apply_observable_kernel (an accumulator per observable)" in eng.describe_schedule()
  bits, params = c3["bits"], c3["params"]
  norm = sum(abs(c) for c, _, _ in whole)
  vals = eng.expectation(bits, params).cpu().numpy()
  assert vals.shape == (len(bits), 3)
  assert np.abs(vals.sum(1) - c3["values"]).max() <= 5e-5 * norm
  w = np.array([[0.7], [-0.4], [0.25], [1.1]], np.float32)
  vals2, grad = eng.expectation_vjp(bits, params, np.repeat(w, 3, axis=1))
  want = (w * c3["grads"]).sum(0)
  tol = 1e-4 * max(1.0, np.abs(want).max())
  assert np.abs(vals2.cpu().numpy() - vals).max() <= 2e-5 * norm
  assert np.abs(grad.cpu().numpy() - want).max() <= tol, np.abs(grad.cpu().numpy() - want).max()
  # unequal weights per observable: the live C oracle
  up3 = np.array([[0.7, -0.2, 0.4], [-0.4, 0.9, 0.1], [0.25, 0.3, -0.8], [1.1, -0.6, 0.5]], np.float32)
  o_vals, o_grad = C.expectation_vjp(n, gates, params, bits, ops3, up3)
  vals3, grad3 = eng.expectation_vjp(bits, params, up3)
  assert np.abs(vals3.cpu().numpy() - o_vals).max() <= 5e-5 * max(sum(abs(c) for c, _, _ in op) for op in ops3)
  assert np.abs(grad3.cpu().numpy() - o_grad).max() <= 1e-4 * max(1.0, np.abs(o_grad).max())
  eng.expectation(bits, params, retain=True)
  if eng.retained is not None:
    grad_r = eng.expectation_vjp_retained(bits, params, up3)
    assert np.abs(grad_r.cpu().numpy() - o_grad).max() <= 1e-4 * max(1.0, np.abs(o_grad).max())


def test_c3_per_state_jacobian_and_shift_rule_against_oracle(c3):
  n, gates, ops = int(c3["n"]), G.gates_of(c3["gates"]), G.ops_of(c3["ops"])
  eng = _engine(n, gates, len(c3["params"]), ops)
  _, jac = eng.expectation_jacobian(c3["bits"], c3["params"])
  jac = jac.cpu().numpy()[:, 0, :]
  assert np.abs(jac - c3["grads"]).max() <= 1e-4 * max(1.0, np.abs(c3["grads"]).max())


def test_c3_full_batch_of_4096_states_oracle_rows_and_position_independence(c3):
  """BASELINE config 3 AT ITS FULL SIZE (4096 states in one resident chunk: 64 GiB of psi and lambda): the
  fixture's four bitstrings sit at the head, in the middle and at the tail of the batch among 4084 random ones.
  Their values and per-state gradient rows must equal the oracle's wherever they sit, bit-identically between
  the three positions (a state's result does not depend on its neighbours, its XCD or its place in the chunk);
  the [P] gradient of the call is the upstream-weighted sum of the rows; a 5-state call returns the same bits."""
  n, gates, ops = int(c3["n"]), G.gates_of(c3["gates"]), G.ops_of(c3["ops"])
  norm = sum(abs(c) for c, _, _ in ops[0])
  eng = _engine(n, gates, len(c3["params"]), ops)
  free, _ = torch.cuda.mem_get_info(0)
  if free < 80 * 2**30:
    pytest.skip("needs 64 GiB of workspace for the resident 4096-state chunk")
  U = 4096
  rng = np.random.default_rng(2024)
  bits = rng.integers(0, 2, size=(U, n)).astype(np.int8)
  places = [0, 2046, U - 4]
  for p0 in places:
    bits[p0:p0 + 4] = c3["bits"]
  up = rng.uniform(0.5, 1.5, size=(U, 1)).astype(np.float32) / U
  vals, grad = eng.expectation_vjp(bits, c3["params"], up)
  rows = eng.state_gradients(U)
  vals, grad, rows = vals.cpu().numpy()[:, 0], grad.cpu().numpy(), rows.cpu().numpy().astype(np.float64)
  tol_g = 1e-4 * max(1.0, np.abs(c3["grads"]).max())
  for p0 in places:
    assert np.abs(vals[p0:p0 + 4] - c3["values"]).max() <= 5e-5 * norm
    got = rows[p0:p0 + 4] / up[p0:p0 + 4].astype(np.float64)
    assert np.abs(got - c3["grads"]).max() <= tol_g, (p0, np.abs(got - c3["grads"]).max())
    assert np.array_equal(vals[p0:p0 + 4], vals[0:4])
  # rows carry the upstream weight: compare the un-weighted rows bit for bit through a call with equal weights
  flat = np.full((U, 1), 1.0 / U, np.float32)
  vals_f, grad_f = eng.expectation_vjp(bits, c3["params"], flat)
  rows_f = eng.state_gradients(U).cpu().numpy()
  for p0 in places[1:]:
    assert np.array_equal(rows_f[p0:p0 + 4], rows_f[0:4])
  assert np.array_equal(vals_f.cpu().numpy()[:, 0], vals)
  assert np.abs(rows_f.astype(np.float64).sum(0) - grad_f.cpu().numpy()).max() <= 1e-5 * max(1.0, np.abs(grad_f.cpu().numpy()).max())
  assert np.abs(rows.sum(0) - grad).max() <= 1e-5 * max(1.0, np.abs(grad).max())
  # the same five states alone
  few = np.concatenate([bits[0:4], bits[1000:1001]])
  v5, _ = eng.expectation_vjp(few, c3["params"], np.full((5, 1), 1.0 / U, np.float32))
  r5 = eng.state_gradients(5).cpu().numpy()
  assert np.array_equal(v5.cpu().numpy()[:, 0], np.concatenate([vals[0:4], vals[1000:1001]]))
  assert np.array_equal(r5, np.concatenate([rows_f[0:4], rows_f[1000:1001]]))


def test_c3_kobe2_shards_through_the_modular_hamiltonian_circuit(c3):
  """bit . U(phi) . V(phi_h)^dagger (1888 gates) measured in the 210 Z-string shards of a KOBE-2
  energy (qnn.py:120-127): all of them come out of one measurement pass."""
  n, gates, shards = int(c3["n"]), G.gates_of(c3["total_gates"]), G.ops_of(c3["kobe2_shards"])
  s = int(c3["kobe2_state"])
  eng = _engine(n, gates, len(c3["total_params"]), shards)
  vals = eng.expectation(c3["bits"][s:s + 1], c3["total_params"]).cpu().numpy()[0]
  assert vals.shape == (210,)
  assert np.abs(vals - c3["kobe2_values"]).max() <= 5e-5, np.abs(vals - c3["kobe2_values"]).max()


def test_c4_all_512_terms_at_24_qubits_against_oracle():
  """BASELINE config 4's observable (random 512-term Pauli sum, 24 qubits), every term its own
  op, HEA depth 2; and the adjoint VJP of the whole sum."""
  g = G.load("c4_n24_d2.npz")
  n, gates, (op,) = int(g["n"]), G.gates_of(g["gates"]), G.ops_of(g["ops"])
  assert len(op) == 512
  eng = _engine(n, gates, len(g["params"]), [[t] for t in op])
  vals = eng.expectation(g["bits"], g["params"]).cpu().numpy()[0]
  coeff = np.array([abs(c) for c, _, _ in op])
  err = np.abs(vals - g["term_values"])
  assert (err <= 5e-5 * np.maximum(1.0, coeff)).all(), err.max()
  eng = _engine(n, gates, len(g["params"]), [op])
  total, grad = eng.expectation_vjp(g["bits"], g["params"], np.ones((1, 1), np.float32))
  assert abs(float(total[0, 0]) - g["term_values"].sum()) <= 5e-5 * coeff.sum()
  assert np.abs(grad.cpu().numpy() - g["grad"]).max() <= 1e-4 * max(1.0, np.abs(g["grad"]).max())


def test_c5_28_qubit_forward_against_oracle():
  """One 2 GiB state vector: TFIM ring terms after a depth-2 HEA at 28 qubits (config 5's width)."""
  free, _ = torch.cuda.mem_get_info()
  if free < 8 << 30:
    pytest.skip("needs 8 GiB of free HBM")
  g = G.load("c5_n28_d2.npz")
  n, gates, (op,) = int(g["n"]), G.gates_of(g["gates"]), G.ops_of(g["ops"])
  eng = _engine(n, gates, len(g["params"]), [[t] for t in op])
  vals = eng.expectation(g["bits"], g["params"]).cpu().numpy()[0]
  assert np.abs(vals - g["term_values"]).max() <= 5e-5, np.abs(vals - g["term_values"]).max()
  eng = _engine(n, gates, len(g["params"]), [op])
  total = float(eng.expectation(g["bits"], g["params"])[0, 0])
  assert abs(total - g["term_values"].sum()) <= 5e-5 * 56


def test_c4_qmhl_loss_through_the_host_mirror_with_parameter_shift_gradients():
  """BASELINE config 4 as stated -- "QMHL loss, parameter-shift grads" -- at 24 qubits, depth 4 + 4:
  `inference.qmhl(data, qhbm)` (qmhl_loss.py:33-34) through the host mirror with
  `gradient_method=GRAD_PARAMETER_SHIFT` (1136 shifted programs per state, batched by the engine), against
  the C oracle's fixture (tests/golden/make_golden_large.py c4qmhl): loss, d/dtheta, d/dphi.  The adjoint
  method must give the same numbers.  Tolerances: loss 5e-5 * sum|theta|; gradients 1e-4 * max(1, |g|_inf)."""
  from qhbmlib_amd import data, inference, ir, models
  from tests.test_host_api import hea_circuit
  g = G.load("c4_qmhl_n24_d4.npz")
  n, layers = int(g["n"]), int(g["layers"])
  qubits = ir.GridQubit.rect(1, n)

  class FixedData(data.QuantumData):
    """Data given as bitstring samples through a fixed circuit (qmhl_loss_test.py:206-215's pattern)."""

    def __init__(self, samples, q_infer):
      self.samples, self.q_infer = samples, q_infer

    def expectation(self, observable):
      return torch.mean(self.q_infer.expectation(self.samples, observable))

  def run(method):
    energy = models.BernoulliEnergy(list(range(n)))
    with torch.no_grad():
      energy.post_process[0].kernel.copy_(torch.as_tensor(g["thetas"], dtype=torch.float32))
    model_circuit = models.DirectQuantumCircuit(hea_circuit(qubits, layers, "m"))
    data_circuit = models.DirectQuantumCircuit(hea_circuit(qubits, layers, "d"))
    with torch.no_grad():
      model_circuit.trainable_variables[0].copy_(torch.as_tensor(g["model_params"], dtype=torch.float32))
      data_circuit.trainable_variables[0].copy_(torch.as_tensor(g["data_params"], dtype=torch.float32))
    # the oracle's circuits are these circuits (same sorted-symbol parameter layout)
    assert model_circuit.pqc.flat_gates(qubits, model_circuit.symbol_names) == G.gates_of(g["model_gates"])
    assert data_circuit.pqc.flat_gates(qubits, data_circuit.symbol_names) == G.gates_of(g["data_gates"])
    qhbm = inference.QHBM(inference.BernoulliEnergyInference(energy, 16, initial_seed=1),
                          inference.AnalyticQuantumInference(model_circuit))
    data_q = inference.AnalyticQuantumInference(data_circuit, gradient_method=method)
    loss = inference.qmhl(FixedData(torch.from_numpy(g["samples"]), data_q), qhbm)
    g_theta, g_phi = torch.autograd.grad(loss, (energy.trainable_variables[0], model_circuit.trainable_variables[0]))
    return float(loss), g_theta.cpu().numpy(), g_phi.cpu().numpy()

  norm = float(np.abs(g["thetas"]).sum())
  for method in (E.GRAD_PARAMETER_SHIFT, E.GRAD_ADJOINT):
    loss, g_theta, g_phi = run(method)
    assert abs(loss - float(g["loss"])) <= 5e-5 * norm, (method, loss, float(g["loss"]))
    np.testing.assert_allclose(g_theta, g["grad_thetas"], atol=1e-4, rtol=0)
    tol = 1e-4 * max(1.0, np.abs(g["grad_model_params"]).max())
    assert np.abs(g_phi - g["grad_model_params"]).max() <= tol, (method, np.abs(g_phi - g["grad_model_params"]).max())
  assert np.abs(g["grad_model_params"]).max() > 1e-2


def test_c4_depth16_parameter_shift_equals_adjoint_on_every_parameter():
  """24 qubits at config 4's full depth 16: the engine's parameter-shift VJP (2 x 1136 shifted programs,
  batched) against its adjoint VJP on ALL 1136 parameters, two states, XXZ + the first 24 terms of the
  random Pauli sum (the adjoint itself is pinned to the oracle at this width in
  test_c4_all_512_terms_at_24_qubits_against_oracle and by the QMHL fixture above)."""
  import bench
  n, layers = 24, 16
  gates, n_params = bench.hea_gates(n, layers)
  ops = [bench.xxz_op(n), bench.random_pauli_op(n, 512, 24)[:24]]
  params = np.random.default_rng(2416).uniform(-1, 1, n_params).astype(np.float32)
  bits = bench.distinct_bitstrings(n, 2, 44)
  eng = _engine(n, gates, n_params, ops)
  up = np.array([[0.8, -0.3], [-0.5, 0.6]], np.float32)
  vals_a, g_adj = eng.expectation_vjp(bits, params, up)
  vals_s, g_shift = eng.expectation_vjp(bits, params, up, method=E.GRAD_PARAMETER_SHIFT)
  g_adj, g_shift = g_adj.cpu().numpy(), g_shift.cpu().numpy()
  np.testing.assert_allclose(vals_s.cpu().numpy(), vals_a.cpu().numpy(), atol=5e-5 * 60)
  assert g_adj.shape == (1136,) and np.abs(g_adj).max() > 1e-2
  assert np.abs(g_shift - g_adj).max() <= 1e-4 * max(1.0, np.abs(g_adj).max()), np.abs(g_shift - g_adj).max()


def test_c4_depth16_parameter_shift_against_the_c_oracle():
  """BASELINE config 4 at its STATED combination -- 24 qubits, depth 16, parameter-shift gradients, terms of the random
  512-term sum -- against the C oracle: the engine's shift VJP (2 x 1136 shifted programs) of the sum
  of the first 24 terms on ALL 1136 parameters, and the 24 term values, against tests/golden/c4_n24_d16_shift.npz
  (make_golden_large.py c4d16: oracle/qhbm_cpu.c adjoint, an independent method on an independent code path).
  Tolerances: values 5e-5 * max(1, |c|); gradient 1e-4 * max(1, |g|_inf)."""
  g = G.load("c4_n24_d16_shift.npz")
  n, gates, (op,) = int(g["n"]), G.gates_of(g["gates"]), G.ops_of(g["ops"])
  assert n == 24 and len(op) == 24 and len(g["params"]) == 1136
  eng = _engine(n, gates, len(g["params"]), [[t] for t in op])
  vals = eng.expectation(g["bits"], g["params"]).cpu().numpy()[0]
  coeff = np.array([abs(c) for c, _, _ in op])
  assert (np.abs(vals - g["term_values"]) <= 5e-5 * np.maximum(1.0, coeff)).all(), np.abs(vals - g["term_values"]).max()
  eng = _engine(n, gates, len(g["params"]), [op])
  total, grad = eng.expectation_vjp(g["bits"], g["params"], np.ones((1, 1), np.float32), method=E.GRAD_PARAMETER_SHIFT)
  grad = grad.cpu().numpy()
  assert abs(float(total[0, 0]) - g["term_values"].sum()) <= 5e-5 * coeff.sum()
  tol = 1e-4 * max(1.0, np.abs(g["grad"]).max())
  assert np.abs(g["grad"]).max() > 1e-2 and np.count_nonzero(np.abs(g["grad"]) > 1e-3) >= 8
  assert np.abs(grad - g["grad"]).max() <= tol, (np.abs(grad - g["grad"]).max(), tol)


def test_c5_streamed_batch_of_three_28_qubit_states_values_and_vjp_against_oracle():
  """Config 5's point is streaming (2 GiB per state, "256 samples, 32 per GPU streamed"): with
  chunk_states = 1 the three states of the fixture go through the workspace one after the other --
  forward (all 56 TFIM terms), and values + adjoint VJP with per-state weights -- and must match the C
  oracle (tests/golden/make_golden_large.py c5vjp) and the unchunked run bit for bit."""
  free, _ = torch.cuda.mem_get_info()
  if free < 24 << 30:
    pytest.skip("needs 24 GiB of free HBM")
  g = G.load("c5_n28_d2_vjp.npz")
  n, gates, (op,) = int(g["n"]), G.gates_of(g["gates"]), G.ops_of(g["ops"])
  bits, params, up = g["bits"], g["params"], g["upstream"].astype(np.float32)
  assert bits.shape == (3, 28)
  eng = _engine(n, gates, len(params), [[t] for t in op], chunk_states=1)
  assert eng.workspace_bytes(3) < 3 * (8 << 28)            # one 2 GiB state at a time
  vals = eng.expectation(bits, params).cpu().numpy()
  assert np.abs(vals - g["term_values"]).max() <= 5e-5, np.abs(vals - g["term_values"]).max()
  del eng
  torch.cuda.empty_cache()
  want_total = g["term_values"].sum(1)
  tol_g = 1e-4 * max(1.0, np.abs(g["grad"]).max())
  results = []
  for chunk in (1, 0):
    eng = _engine(n, gates, len(params), [op], **({"chunk_states": chunk} if chunk else {}))
    total, grad = eng.expectation_vjp(bits, params, up)
    total, grad = total.cpu().numpy()[:, 0], grad.cpu().numpy()
    assert np.abs(total - want_total).max() <= 5e-5 * 56, np.abs(total - want_total).max()
    assert np.abs(grad - g["grad"]).max() <= tol_g, np.abs(grad - g["grad"]).max()
    results.append((total, grad))
    del eng
    torch.cuda.empty_cache()
  np.testing.assert_array_equal(results[0][0], results[1][0])      # streaming changes no bit
  np.testing.assert_array_equal(results[0][1], results[1][1])
