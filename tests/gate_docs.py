"""cirq 0.14.1's DOCUMENTED matrices of the twelve power-gate kinds, written out entry by entry.

These are the closed forms printed in the docstrings of cirq/ops/common_gates.py, swap_gates.py and
parity_gates.py (XPowGate, YPowGate, ZPowGate, HPowGate, CZPowGate, CXPowGate, SwapPowGate, ISwapPowGate,
XXPowGate, YYPowGate, ZZPowGate) and of cirq.rx / ry / rz -- NOT the eigen-decomposition the oracle is built
from (oracle/qhbm_oracle.py::_eigen_components), so that comparing the two pins the oracle's gate table to the
published convention, and comparing the engine with them pins every gate kind end to end.
First qubit = most significant bit of the matrix index (cirq's convention).
"""
import math

import numpy as np

from oracle import qhbm_oracle as O


def documented_matrix(kind, t):
  c, s = math.cos(math.pi * t / 2), math.sin(math.pi * t / 2)
  g = np.exp(1j * math.pi * t / 2)
  e = np.exp(1j * math.pi * t)
  r2 = math.sqrt(2.0)
  if kind == O.GATE_I:
    return np.eye(2, dtype=complex)
  if kind == O.GATE_XPOW:    # "X**t = [[g c, -i g s], [-i g s, g c]]"
    return np.array([[g * c, -1j * g * s], [-1j * g * s, g * c]])
  if kind == O.GATE_YPOW:    # "Y**t = [[g c, -g s], [g s, g c]]"
    return np.array([[g * c, -g * s], [g * s, g * c]])
  if kind == O.GATE_ZPOW:    # "Z**t = [[1, 0], [0, g^2]]"
    return np.array([[1, 0], [0, e]])
  if kind == O.GATE_HPOW:    # "H**t = g [[c - i s / sqrt 2, -i s / sqrt 2], [-i s / sqrt 2, c + i s / sqrt 2]]"
    return g * np.array([[c - 1j * s / r2, -1j * s / r2], [-1j * s / r2, c + 1j * s / r2]])
  if kind == O.GATE_CZPOW:   # "CZ**t = diag(1, 1, 1, g^2)"
    return np.diag([1, 1, 1, e])
  if kind == O.GATE_CNOTPOW:  # "CNOT**t = [[1,0,0,0],[0,1,0,0],[0,0,g c,-i g s],[0,0,-i g s,g c]]" (control first)
    return np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, g * c, -1j * g * s], [0, 0, -1j * g * s, g * c]])
  if kind == O.GATE_SWAPPOW:  # "SWAP**t = [[1,0,0,0],[0,g c,-i g s,0],[0,-i g s,g c,0],[0,0,0,1]]"
    return np.array([[1, 0, 0, 0], [0, g * c, -1j * g * s, 0], [0, -1j * g * s, g * c, 0], [0, 0, 0, 1]])
  if kind == O.GATE_ISWAPPOW:  # "ISWAP**t = [[1,0,0,0],[0,c,i s,0],[0,i s,c,0],[0,0,0,1]]"
    return np.array([[1, 0, 0, 0], [0, c, 1j * s, 0], [0, 1j * s, c, 0], [0, 0, 0, 1]])
  cc, ss = g * c, -1j * g * s
  if kind == O.GATE_XXPOW:   # "XX**t = [[c,0,0,s],[0,c,s,0],[0,s,c,0],[s,0,0,c]], c = f cos, s = -i f sin, f = e^{i pi t/2}"
    return np.array([[cc, 0, 0, ss], [0, cc, ss, 0], [0, ss, cc, 0], [ss, 0, 0, cc]])
  if kind == O.GATE_YYPOW:   # "YY**t = [[c,0,0,-s],[0,c,s,0],[0,s,c,0],[-s,0,0,c]]"
    return np.array([[cc, 0, 0, -ss], [0, cc, ss, 0], [0, ss, cc, 0], [-ss, 0, 0, cc]])
  if kind == O.GATE_ZZPOW:   # "ZZ**t = diag(1, w, w, 1), w = e^{i pi t}"
    return np.diag([1, e, e, 1])
  raise ValueError(kind)


def documented_rotation(axis, theta):
  """cirq.rx / ry / rz(theta) = exp(-i theta P / 2): the global_shift = -1/2 forms of X / Y / ZPowGate(theta / pi)."""
  c, s = math.cos(theta / 2), math.sin(theta / 2)
  if axis == "x":
    return np.array([[c, -1j * s], [-1j * s, c]])
  if axis == "y":
    return np.array([[c, -s], [s, c]])
  return np.diag([np.exp(-1j * theta / 2), np.exp(1j * theta / 2)])


PAULI = {"I": np.eye(2, dtype=complex), "X": np.array([[0, 1], [1, 0]], dtype=complex),
         "Y": np.array([[0, -1j], [1j, 0]], dtype=complex), "Z": np.diag([1.0 + 0j, -1.0])}


def probe_values(kind, t, bits, probes, paulis):
  """<x| P^dag G(t)^dag O G(t) P |x> from the DOCUMENTED matrices only: P = the product of X**probes[q] on the gate's
  qubits (X**p is the one gate the reference's own tests pin, qnn_test.py:83-180), O each Pauli string of `paulis`."""
  nq = 1 if kind in (O.GATE_I, O.GATE_XPOW, O.GATE_YPOW, O.GATE_ZPOW, O.GATE_HPOW) else 2
  prep = documented_matrix(O.GATE_XPOW, probes[0])
  if nq == 2:
    prep = np.kron(prep, documented_matrix(O.GATE_XPOW, probes[1]))
  idx = 0
  for b in bits[:nq]:
    idx = 2 * idx + int(b)
  psi = np.zeros(2**nq, dtype=complex)
  psi[idx] = 1.0
  psi = documented_matrix(kind, t) @ (prep @ psi)
  out = []
  for string in paulis:
    op = PAULI[string[0]]
    if nq == 2:
      op = np.kron(op, PAULI[string[1]])
    out.append(float(np.real(np.vdot(psi, op @ psi))))
  return np.array(out)
