"""The plugin boundary of the host mirror on CPU (no GPU, no engine compute):

* `QuantumInference._expectation` keeps the reference's signature
  (`/root/reference/qhbmlib/inference/qnn.py:82-84`: circuits, symbol_names, symbol_values,
  observables), so a subclass written against the reference ports unchanged;
* `AnalyticQuantumInference` keys its engines on CONTENT (VERDICT r1 weak #1 / ADVICE r1): fresh
  operator lists that reuse a freed list's `id()` must reach an engine holding THEIR observables.
"""
import itertools

import numpy as np
import pytest
import torch

from qhbmlib_amd import inference, ir, models
from qhbmlib_amd.inference import qnn as qnn_module
from tests.test_host_api import hea_circuit


def _set(param, values):
  with torch.no_grad():
    param.copy_(torch.as_tensor(np.asarray(values), dtype=torch.float32))


def test_reference_signature_subclass_is_driven_like_the_reference():
  """A plugin in the reference's shape: receives the resolved circuits (one per UNIQUE bitstring),
  the [P] symbol names, the [U, P] tiled symbol values and the observables untouched; its [U, T]
  result is expanded back to input row order (qnn.py:68-80)."""
  n = 3
  qubits = ir.GridQubit.rect(1, n)
  circ = models.DirectQuantumCircuit(hea_circuit(qubits, 1, "s"))
  vals = np.linspace(-0.5, 0.5, len(circ.symbol_names)).astype(np.float32)
  _set(circ.trainable_variables[0], vals)
  seen = {}

  class Plugin(inference.QuantumInference):

    def _expectation(self, circuits, symbol_names, symbol_values, observables):
      seen.update(circuits=circuits, symbol_names=symbol_names, symbol_values=symbol_values,
                  observables=observables)
      bits, total = circuits
      assert total.symbol_names == list(symbol_names)
      # row u: (integer value of bitstring u) + sum of the tiled parameters of row u
      ints = (bits.to(torch.float32) * torch.tensor([4.0, 2.0, 1.0])).sum(1, keepdim=True)
      n_ops = len(observables) if isinstance(observables, list) else 1
      return (ints + symbol_values.sum(1, keepdim=True)).repeat(1, n_ops)

  ops = [ir.PZ(qubits[0]) + ir.PZ(qubits[1]), 1.0 * ir.PX(qubits[2])]
  states = torch.tensor([[1, 0, 1], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 0, 1]], dtype=torch.int8)
  plugin = Plugin(circ)
  out = plugin.expectation(states, ops)
  assert out.shape == (5, 2)
  np.testing.assert_allclose(out.detach().numpy()[:, 0], np.array([5, 1, 5, 7, 1]) + vals.sum(), rtol=1e-6)
  bits, total = seen["circuits"]
  assert bits.tolist() == [[1, 0, 1], [0, 0, 1], [1, 1, 1]]   # unique, first-occurrence order (Q4)
  assert seen["circuits"].num_circuits == 3 and total is circ
  assert seen["symbol_names"] == circ.symbol_names
  assert tuple(seen["symbol_values"].shape) == (3, len(circ.symbol_names))
  np.testing.assert_allclose(seen["symbol_values"].detach().numpy(), np.tile(vals, (3, 1)))
  assert seen["observables"] is ops
  # differentiable through the tile, as tf.tile is at qnn.py:75-76: d/d phi_p = sum over rows and ops
  (g,) = torch.autograd.grad(out.sum(), circ.trainable_variables)
  np.testing.assert_allclose(g.numpy(), np.full(len(vals), 5 * 2.0))
  # the Hamiltonian branch appends the inverse circuit before resolving (qnn.py:69-72)
  ham = models.Hamiltonian(models.BernoulliEnergy(list(range(n))),
                           models.DirectQuantumCircuit(hea_circuit(qubits, 1, "h")))
  plugin.expectation(states, ham)
  _, total = seen["circuits"]
  assert len(total.pqc) == 2 * len(circ.pqc) and total.symbol_names == seen["symbol_names"]
  assert seen["observables"] is ham


class _StubEngine:
  """Stands in for `_engine.Engine` on a CPU box: remembers what it was given and returns, for
  every state and op, the SUM OF THE OP'S z-MASKS -- enough to tell which observables an engine
  was built with."""
  created = 0

  def __init__(self, device):
    type(self).created += 1
    self.device, self.retained, self.ops = device, None, None

  def set_circuit(self, n, gates, n_params):
    self.circuit = (n, tuple(gates), n_params)

  def set_observables(self, ops):
    self.ops = ops

  def set_gradient_mask(self, needs_grad):
    self.gradient_mask = needs_grad

  def allocated_bytes(self):
    return 1 << 20

  def expectation(self, bits, params, retain=False):
    row = torch.tensor([float(sum(z for _, _, z in op)) for op in self.ops])
    return row.unsqueeze(0).repeat(bits.shape[0], 1)


@pytest.fixture
def stub_engine(monkeypatch):
  _StubEngine.created = 0
  monkeypatch.setattr(qnn_module._engine, "Engine", _StubEngine)
  monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
  monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
  return _StubEngine


def test_engine_cache_is_keyed_on_content_not_identity(stub_engine):
  n = 3
  qubits = ir.GridQubit.rect(1, n)
  circ = models.DirectQuantumCircuit(hea_circuit(qubits, 1, "k"))
  qnn = inference.AnalyticQuantumInference(circ)
  states = torch.tensor(list(itertools.product([0, 1], repeat=n)), dtype=torch.int8)
  collisions = 0
  last_ids = None
  for _ in range(6):
    for k, pauli in enumerate((ir.PX, ir.PY, ir.PZ)):
      ops = [1.0 * pauli(q) for q in qubits]          # a fresh list every time
      ids = tuple(id(o) for o in ops)
      collisions += ids == last_ids
      last_ids = ids
      got = qnn.expectation(states, ops).detach().numpy()
      want = [0, 0, 0] if k == 0 else [1, 2, 4]       # z-masks: X has none, Y and Z have the qubit's bit
      np.testing.assert_array_equal(got, np.tile(want, (8, 1)))
  # equal content -> same engine: X lists share one, Y and Z differ in x-masks -> three engines
  assert stub_engine.created == 3
  # in-place mutation of an operator changes the key
  op = ir.PauliSum.from_pauli_strings([ir.PZ(qubits[0])])
  assert qnn.expectation(states, [op])[0, 0].item() == 1.0
  op += ir.PZ(qubits[2])
  assert qnn.expectation(states, [op])[0, 0].item() == 5.0
  # a different circuit behind the same observables is a different engine too (Hamiltonian branch)
  before = stub_engine.created
  for layers in (1, 2):
    ham = models.Hamiltonian(models.BernoulliEnergy(list(range(n))),
                             models.DirectQuantumCircuit(hea_circuit(qubits, layers, "h")))
    assert qnn.expectation(states, ham).shape == (8, 1)
  assert stub_engine.created == before + 2


def test_engine_cache_is_bounded(stub_engine):
  n = 2
  qubits = ir.GridQubit.rect(1, n)
  circ = models.DirectQuantumCircuit(hea_circuit(qubits, 1, "b"))
  qnn = inference.AnalyticQuantumInference(circ, max_cached_engines=2)
  states = torch.zeros((1, n), dtype=torch.int8)
  for c in (0.5, 1.5, 2.5, 3.5):
    qnn.expectation(states, [c * ir.PZ(qubits[0])])
  assert len(qnn._engines) == 2 and stub_engine.created == 4
  qnn.expectation(states, [3.5 * ir.PZ(qubits[0])])   # most recent: still cached
  assert stub_engine.created == 4
  qnn.expectation(states, [0.5 * ir.PZ(qubits[0])])   # evicted: rebuilt
  assert stub_engine.created == 5


def test_rows_of_symbol_values_must_agree():
  n = 2
  qubits = ir.GridQubit.rect(1, n)
  circ = models.DirectQuantumCircuit(hea_circuit(qubits, 1, "r"))
  qnn = inference.AnalyticQuantumInference(circ)
  bits = torch.zeros((2, n), dtype=torch.int8)
  vals = torch.stack([circ.symbol_values, circ.symbol_values + 1.0])
  with pytest.raises(ValueError, match="rows of symbol_values differ"):
    qnn._expectation(qnn_module.ResolvedCircuits(bits, circ), circ.symbol_names, vals, [ir.PZ(qubits[0])])
