"""Inference on quantum circuits (reference: qhbmlib/inference/qnn.py).

`AnalyticQuantumInference` is the drop-in for the reference class of the same
name: where the reference tiles protos and calls `tfq.layers.Expectation`
(qnn.py:112,133-138), this one hands the unique bitstrings and the current
symbol values to the HIP engine through the C ABI.  There is no CPU path.
"""
import abc
from typing import List, Sequence, Union

import torch

from qhbmlib_amd import _engine
from qhbmlib_amd import ir
from qhbmlib_amd import utils
from qhbmlib_amd.models import circuit  # noqa: F401
from qhbmlib_amd.models import energy
from qhbmlib_amd.models import hamiltonian

Observables = Union[Sequence[ir.PauliSumLike], hamiltonian.Hamiltonian]


class _ExpectationFunction(torch.autograd.Function):
  """values[U, T] = engine(bits, symbol_values); backward = adjoint VJP in the engine
  (the role TFQ's adjoint differentiator plays at qnn.py:90-99)."""

  @staticmethod
  def forward(ctx, symbol_values, engine, bits, method):
    ctx.engine, ctx.bits, ctx.method = engine, bits, method
    ctx.save_for_backward(symbol_values)
    return engine.expectation(bits, symbol_values.detach())

  @staticmethod
  def backward(ctx, upstream):
    (symbol_values,) = ctx.saved_tensors
    _, grad = ctx.engine.expectation_vjp(ctx.bits, symbol_values.detach(), upstream.contiguous(),
                                         ctx.method)
    return grad.to(symbol_values.device), None, None, None


class QuantumInference(torch.nn.Module, abc.ABC):
  """Interface for inference on quantum circuits (qnn.py:29-84)."""

  def __init__(self, input_circuit: circuit.QuantumCircuit, name: Union[None, str] = None):
    super().__init__()
    self.name = name or type(self).__name__
    input_circuit.build([])
    self._circuit = input_circuit

  @property
  def circuit(self):
    return self._circuit

  def expectation(self, initial_states: torch.Tensor, observables: Observables):
    """[batch_size, n_ops] of <x_i| C^dagger O_j C |x_i>, rows in input order
    (qnn.py:50-80).  `observables` is a list of PauliSums or a Hamiltonian."""
    initial_states = torch.as_tensor(initial_states)
    unique_states, idx, _ = utils.unique_bitstrings_with_counts(initial_states)
    if isinstance(observables, hamiltonian.Hamiltonian):
      total_circuit = self.circuit + observables.circuit_dagger
    else:
      total_circuit = self.circuit
    unique_expectations = self._expectation(unique_states, total_circuit, observables)
    return utils.expand_unique_results(unique_expectations, idx)

  @abc.abstractmethod
  def _expectation(self, unique_states, total_circuit, observables):
    raise NotImplementedError()


class AnalyticQuantumInference(QuantumInference):
  """Exact expectation values with adjoint gradients on the MI355X engine
  (qnn.py:87-139).  `gradient_method` may be set to
  `_engine.GRAD_PARAMETER_SHIFT` to use two shifted forwards per gate occurrence
  instead (the rule of tfq.differentiators.ParameterShift, qnn.py:168)."""

  def __init__(self, input_circuit: circuit.QuantumCircuit, name: Union[None, str] = None,
               device: Union[None, int] = None, gradient_method: int = _engine.GRAD_ADJOINT):
    super().__init__(input_circuit, name)
    self._device = device
    self.gradient_method = gradient_method
    self._engines = {}

  def _engine_for(self, total_circuit, ops: List[ir.PauliSum], key):
    cached = self._engines.get(key)
    if cached is not None:
      return cached
    if not torch.cuda.is_available():
      raise _engine.EngineError(
          "AnalyticQuantumInference needs an MI355X: the expectation engine is HIP-only "
          "and has no CPU fallback")
    dev = torch.cuda.current_device() if self._device is None else self._device
    eng = _engine.Engine(dev)
    qubits = total_circuit.qubits
    eng.set_circuit(len(qubits), total_circuit.pqc.flat_gates(qubits, total_circuit.symbol_names),
                    len(total_circuit.symbol_names))
    eng.set_observables([ir.as_pauli_sum(op).masks(qubits) for op in ops])
    self._engines[key] = eng
    return eng

  def _expectation(self, unique_states, total_circuit, observables):
    """See qnn.py:114-139.  A Hamiltonian is only accepted if its energy inherits from
    PauliMixin."""
    if isinstance(observables, hamiltonian.Hamiltonian):
      if not isinstance(observables.energy, energy.PauliMixin):
        raise TypeError("General Hamiltonians not accepted.  "
                        "Please use `SampledQuantumInference` instead.")
      ops = observables.operator_shards
      post_process = lambda y: observables.energy.operator_expectation(y).unsqueeze(-1)
      key = ("hamiltonian", id(observables))
    else:
      ops = list(observables)
      post_process = lambda x: x
      key = ("ops", tuple(id(o) for o in ops))
    eng = self._engine_for(total_circuit, ops, key)
    perm = total_circuit.bit_column_to_qubit()
    bits = unique_states.to(torch.int8)
    if perm != list(range(len(perm))):
      permuted = torch.zeros_like(bits)
      permuted[:, perm] = bits
      bits = permuted
    symbol_values = total_circuit.symbol_values.to(torch.float32)
    expectations = _ExpectationFunction.apply(symbol_values, eng, bits, self.gradient_method)
    return post_process(expectations)
