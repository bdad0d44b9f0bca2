"""Utilities for metrics on QuantumCircuit (reference: qhbmlib/inference/qnn_utils.py)."""
import itertools

import torch

from qhbmlib_amd import _engine
from qhbmlib_amd.models import circuit


def unitary(input_circuit: circuit.QuantumCircuit, device=None):
  """Returns the unitary matrix corresponding to the given circuit (qnn_utils.py:23-33).

  Where the reference calls `tfq.layers.Unitary`, column x of the matrix is the engine's final
  state for basis input |x> (`qhbm_statevector`): one batched forward over all 2^n bitstrings.
  Row/column index = bitstring read big-endian over `sorted(qubits)`, as cirq orders unitaries.
  Not differentiable (the reference's metric code never differentiates it either).
  """
  if not torch.cuda.is_available():
    raise _engine.EngineError("unitary() needs an MI355X: the engine is HIP-only, no CPU fallback")
  input_circuit.build([])
  qubits = input_circuit.qubits
  n = len(qubits)
  eng = _engine.Engine(torch.cuda.current_device() if device is None else device)
  eng.set_circuit(n, input_circuit.pqc.flat_gates(qubits, input_circuit.symbol_names),
                  len(input_circuit.symbol_names))
  bits = torch.tensor(list(itertools.product([0, 1], repeat=n)), dtype=torch.int8)
  states = eng.statevector(bits, input_circuit.symbol_values.detach().to(torch.float32))
  eng.close()
  return states.transpose(0, 1).contiguous()
