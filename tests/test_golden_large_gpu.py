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
                                  {"tile_qubits": 14, "adjoint_tile_qubits": 13}],
                         ids=["default-plan", "chunked", "small-tiles", "large-tiles"])
def test_c3_values_and_vjp_against_oracle(c3, opts):
  """BASELINE config 3's circuit (20 qubits, depth 16, 944 parameters, XXZ): default 9 / 11-pass
  plans, a chunked batch and two other tile geometries, on |0..0>, |1..1> and two random states."""
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


def test_c3_per_state_jacobian_and_shift_rule_against_oracle(c3):
  n, gates, ops = int(c3["n"]), G.gates_of(c3["gates"]), G.ops_of(c3["ops"])
  eng = _engine(n, gates, len(c3["params"]), ops)
  _, jac = eng.expectation_jacobian(c3["bits"], c3["params"])
  jac = jac.cpu().numpy()[:, 0, :]
  assert np.abs(jac - c3["grads"]).max() <= 1e-4 * max(1.0, np.abs(c3["grads"]).max())


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
