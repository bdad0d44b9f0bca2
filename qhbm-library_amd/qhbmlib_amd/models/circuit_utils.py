"""Utilities for the circuit module (reference: qhbmlib/models/circuit_utils.py)."""
from typing import List

from qhbmlib_amd import ir


def bit_circuit(qubits: List[ir.GridQubit], name="bit_circuit"):
  """X**bit on every qubit with symbols `{name}_bit_{n}` (circuit_utils.py:23-29).
  The engine never simulates it: X**1 = X and X**0 = I exactly, so the injector
  is the initial basis state (the `bits` argument of the C ABI)."""
  circuit = ir.Circuit()
  for n, q in enumerate(qubits):
    circuit += ir.X(q)**ir.Symbol(f"{name}_bit_{n}")
  return circuit


def bit_symbol_names(qubits, name="bit_circuit"):
  """sorted() of the injector's symbol strings -- lexicographic, so for n >= 11
  `..._10` sorts before `..._2` (circuit.py:59-62)."""
  return sorted(f"{name}_bit_{n}" for n in range(len(qubits)))


def tfq_bit_permutation(num_qubits):
  """perm[j] = qubit driven by bitstring column j in the reference
  (circuit.py:132-134 assigns column j to the j-th SORTED symbol name)."""
  return [int(s.rsplit("_", 1)[1]) for s in bit_symbol_names(range(num_qubits))]
