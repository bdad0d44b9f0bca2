"""Dense metrics on the engine: statevector, unitary, density_matrix, fidelity.

Reference tests mirrored: tests/inference/qnn_utils_test.py (unitary against the simulator ->
here against the numpy oracle), tests/inference/qhbm_utils_test.py:28-51 (Bell density matrix),
:61-80 (fidelity with itself), :82-118 (fidelity against the direct sqrtm formula).
Tolerances: complex64 amplitudes 2e-6 absolute; fidelity rtol 1e-4 (the reference's close_rtol).
"""
import itertools

import numpy as np
import pytest
import torch

from oracle import qhbm_oracle as O
from qhbmlib_amd import _engine as E
from qhbmlib_amd import inference, ir, models
from tests.test_engine_gpu import random_circuit
from tests.test_host_api import hea_circuit

pytestmark = pytest.mark.gpu


def _set(param, values):
  with torch.no_grad():
    param.copy_(torch.as_tensor(np.asarray(values), dtype=torch.float32))


@pytest.mark.parametrize("n,tile", [(3, 0), (10, 0), (12, 10), (15, 0)])
def test_statevector_matches_oracle(n, tile):
  rng = np.random.default_rng(n)
  n_params = 6
  gates = random_circuit(rng, n, 40, n_params)
  params = rng.uniform(-1, 1, n_params)
  bits = rng.integers(0, 2, size=(3, n)).astype(np.int8)
  eng = E.Engine(0)
  if tile:
    eng.set_option("tile_qubits", tile)
  eng.set_circuit(n, gates, n_params)          # no observables installed
  got = eng.statevector(bits, params).cpu().numpy()
  assert got.shape == (3, 2**n) and got.dtype == np.complex64
  for row, b in zip(got, bits):
    want = O.simulate(n, gates, params, list(b)).ravel()
    np.testing.assert_allclose(row, want, atol=2e-6)
  # with observables installed the state is the same and the values are still right
  op = O.xxz_chain_op(n)
  eng.set_observables([op])
  np.testing.assert_allclose(eng.statevector(bits, params).cpu().numpy(), got, atol=1e-7)
  np.testing.assert_allclose(eng.expectation(bits, params).cpu().numpy(),
                             O.expectation(n, gates, params, bits, [op]), atol=1e-5 * 3 * n)


def test_statevector_empty_batch_and_errors():
  eng = E.Engine(0)
  with pytest.raises(E.EngineError, match="qhbm_set_circuit"):
    eng.n_qubits = 2
    eng.statevector(np.zeros((1, 2), np.int8), np.zeros(0, np.float32))
  eng.set_circuit(2, [(E.GATE_HPOW, 0, -1, -1, 0.0, 1.0)], 0)
  assert eng.statevector(np.zeros((0, 2), np.int8), np.zeros(0, np.float32)).shape == (0, 4)


def test_unitary_matches_oracle():
  n = 4
  qubits = ir.GridQubit.rect(1, n)
  raw = hea_circuit(qubits, 2, "u")
  circ = models.DirectQuantumCircuit(raw)
  values = np.random.default_rng(5).uniform(-1, 1, len(circ.symbol_names))
  _set(circ.trainable_variables[0], values)
  got = inference.unitary(circ).cpu().numpy()
  want = O.unitary(n, raw.flat_gates(circ.qubits, circ.symbol_names), values)
  np.testing.assert_allclose(got, want, atol=2e-6)
  np.testing.assert_allclose(got.conj().T @ got, np.eye(2**n), atol=1e-5)


def test_density_matrix_bell_state():
  """qhbm_utils_test.py:28-51."""
  qubits = ir.GridQubit.rect(1, 2)
  energy = models.BernoulliEnergy([0, 1])
  energy.build([None, 2])
  _set(energy.trainable_variables[0], [-10.0, -10.0])  # pin at |00>
  circ = models.DirectQuantumCircuit(ir.Circuit(ir.H(qubits[0]), ir.CNOT(qubits[0], qubits[1])))
  model = models.Hamiltonian(energy, circ)
  expected = np.array([[0.5, 0, 0, 0.5], [0, 0, 0, 0], [0, 0, 0, 0], [0.5, 0, 0, 0.5]])
  np.testing.assert_allclose(inference.density_matrix(model).detach().cpu().numpy(), expected, atol=1e-6)


def _random_model(n, layers, seed):
  qubits = ir.GridQubit.rect(1, n)
  rng = np.random.default_rng(seed)
  energy = models.KOBE(list(range(n)), n)
  energy.build([None, n])
  thetas = rng.uniform(-1, 1, energy.trainable_variables[0].shape)
  _set(energy.trainable_variables[0], thetas)
  raw = hea_circuit(qubits, layers, "fid")
  circ = models.DirectQuantumCircuit(raw)
  phis = rng.uniform(-1, 1, len(circ.symbol_names))
  _set(circ.trainable_variables[0], phis)
  flat = raw.flat_gates(circ.qubits, circ.symbol_names)
  return models.Hamiltonian(energy, circ), flat, phis, thetas


def test_fidelity_self_and_random():
  """qhbm_utils_test.py:61-118."""
  n = 4
  model, flat, phis, thetas = _random_model(n, 3, 21)
  dm = inference.density_matrix(model).detach()
  want_dm = O.density_matrix(n, flat, phis, lambda b: O.kobe_energy(b, thetas, n))
  np.testing.assert_allclose(dm.cpu().numpy(), want_dm, atol=2e-6)
  np.testing.assert_allclose(float(inference.fidelity(model, dm).detach()), 1.0, rtol=1e-4)
  rng = np.random.default_rng(3)
  for _ in range(3):
    a = rng.normal(size=(2**n, 2**n)) + 1j * rng.normal(size=(2**n, 2**n))
    sigma = a @ a.conj().T
    sigma /= np.trace(sigma).real
    want = O.fidelity_direct(want_dm, sigma)
    got = float(inference.fidelity(model, torch.from_numpy(sigma)).detach())   # complex128 in, cast inside
    np.testing.assert_allclose(got, want, rtol=1e-4)
