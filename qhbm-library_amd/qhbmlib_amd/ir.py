"""Circuit and Pauli-sum IR of the host side.

The reference describes circuits with cirq objects and ships them to the
simulator as serialized protos (`tfq.convert_to_tensor`,
/root/reference/qhbmlib/models/circuit.py:63,171-174,207).  Neither cirq nor
TFQ exists on the MI355X box, so the host keeps a small IR of its own with the
same *semantics*: gates are cirq's one-parameter power gates with exponents of
the form `scalar * symbol + offset` (all TFQ can serialize), a circuit is an
ordered gate list, `circuit ** -1` reverses it and negates every exponent
(circuit.py:164-176).  The IR lowers 1:1 to the flat gate list of the C ABI
(include/qhbm_engine.h).
"""
import dataclasses
import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

from qhbmlib_amd import _engine as E


# ---------------------------------------------------------------------------
# Qubits
# ---------------------------------------------------------------------------
@dataclasses.dataclass(frozen=True, order=True)
class GridQubit:
  """Stand-in for cirq.GridQubit: ordered row-major."""
  row: int
  col: int

  @staticmethod
  def rect(rows: int, cols: int) -> List["GridQubit"]:
    return [GridQubit(r, c) for r in range(rows) for c in range(cols)]

  def __repr__(self):
    return f"q({self.row}, {self.col})"


# ---------------------------------------------------------------------------
# Exponents:  scalar * symbol + offset
# ---------------------------------------------------------------------------
@dataclasses.dataclass(frozen=True)
class Symbol:
  """Stand-in for sympy.Symbol; supports `c * s`, `s * c`, `s / c`, `-s`, `s + c`."""
  name: str

  def _expr(self):
    return Exponent(self.name, 1.0, 0.0)

  def __mul__(self, c):
    return self._expr() * c

  __rmul__ = __mul__

  def __truediv__(self, c):
    return self._expr() * (1.0 / c)

  def __neg__(self):
    return self._expr() * -1.0

  def __add__(self, c):
    return self._expr() + c

  __radd__ = __add__

  def __sub__(self, c):
    return self._expr() + (-c)

  def __str__(self):
    return self.name


@dataclasses.dataclass(frozen=True)
class Exponent:
  symbol: Optional[str]
  scalar: float
  offset: float

  def __mul__(self, c):
    return Exponent(self.symbol, self.scalar * float(c), self.offset * float(c))

  __rmul__ = __mul__

  def __truediv__(self, c):
    return self * (1.0 / float(c))

  def __neg__(self):
    return self * -1.0

  def __add__(self, c):
    return Exponent(self.symbol, self.scalar, self.offset + float(c))

  __radd__ = __add__

  def __sub__(self, c):
    return self + (-float(c))


def symbols(names: str) -> Tuple[Symbol, ...]:
  """`sympy.symbols("a b c")`."""
  return tuple(Symbol(s) for s in names.replace(",", " ").split())


def _as_exponent(x) -> Exponent:
  if isinstance(x, Exponent):
    return x
  if isinstance(x, Symbol):
    return x._expr()  # pylint: disable=protected-access
  return Exponent(None, 0.0, float(x))


# ---------------------------------------------------------------------------
# Gates
# ---------------------------------------------------------------------------
_KIND_NAMES = {
    E.GATE_I: "I", E.GATE_XPOW: "X", E.GATE_YPOW: "Y", E.GATE_ZPOW: "Z",
    E.GATE_HPOW: "H", E.GATE_CZPOW: "CZ", E.GATE_CNOTPOW: "CNOT",
    E.GATE_SWAPPOW: "SWAP", E.GATE_ISWAPPOW: "ISWAP", E.GATE_XXPOW: "XX",
    E.GATE_YYPOW: "YY", E.GATE_ZZPOW: "ZZ",
}


@dataclasses.dataclass(frozen=True)
class Gate:
  """exp(i pi t global_shift) * kind ** t on `qubits`, t = `exponent` (cirq EigenGate convention).
  The global phase never changes an expectation value, a gradient or a sample; the exported
  statevector and `inference.unitary` carry it (cirq.rx/ry/rz: global_shift = -0.5)."""
  kind: int
  qubits: Tuple[GridQubit, ...]
  exponent: Exponent = Exponent(None, 0.0, 1.0)
  global_shift: float = 0.0

  def __pow__(self, power):
    if isinstance(power, (Symbol, Exponent)):
      base = self.exponent
      if base.symbol is not None:
        raise ValueError("cannot raise a symbolic gate to a symbolic power")
      return Gate(self.kind, self.qubits, _as_exponent(power) * base.offset, self.global_shift)
    return Gate(self.kind, self.qubits, self.exponent * float(power), self.global_shift)

  def inverse(self) -> "Gate":
    # (e^{i pi t g} G^t)^-1 = e^{i pi (-t) g} G^{-t}: the exponent changes sign, the shift stays
    return Gate(self.kind, self.qubits, -self.exponent, self.global_shift)

  def __repr__(self):
    e = self.exponent
    ex = (f"{e.scalar:g}*{e.symbol}" if e.symbol else "") + (f"{e.offset:+g}" if e.offset or not e.symbol else "")
    gs = f", global_shift={self.global_shift:g}" if self.global_shift else ""
    return f"{_KIND_NAMES[self.kind]}{list(self.qubits)}**({ex}{gs})"


def _g1(kind):
  return lambda q: Gate(kind, (q,))


def _g2(kind):
  return lambda q0, q1: Gate(kind, (q0, q1))


I = _g1(E.GATE_I)
X = _g1(E.GATE_XPOW)
Y = _g1(E.GATE_YPOW)
Z = _g1(E.GATE_ZPOW)
H = _g1(E.GATE_HPOW)
CZ = _g2(E.GATE_CZPOW)
CNOT = _g2(E.GATE_CNOTPOW)
SWAP = _g2(E.GATE_SWAPPOW)
ISWAP = _g2(E.GATE_ISWAPPOW)
XX = _g2(E.GATE_XXPOW)
YY = _g2(E.GATE_YYPOW)
ZZ = _g2(E.GATE_ZZPOW)


def CZPowGate(exponent):  # pylint: disable=invalid-name
  """`cirq.CZPowGate(exponent=a)(q0, q1)` (tests/test_util.py:30-32)."""
  return lambda q0, q1: CZ(q0, q1)**exponent


def _rotation(kind, theta):
  return lambda q: Gate(kind, (q,), _as_exponent(theta) * (1.0 / math.pi), -0.5)


def rx(theta):
  """cirq.rx(theta) = XPowGate(exponent=theta/pi, global_shift=-0.5) = exp(-i theta X / 2)."""
  return _rotation(E.GATE_XPOW, theta)


def ry(theta):
  """cirq.ry(theta) = YPowGate(exponent=theta/pi, global_shift=-0.5) = exp(-i theta Y / 2)."""
  return _rotation(E.GATE_YPOW, theta)


def rz(theta):
  """cirq.rz(theta) = ZPowGate(exponent=theta/pi, global_shift=-0.5) = exp(-i theta Z / 2)."""
  return _rotation(E.GATE_ZPOW, theta)


def phased_x_pow(q, phase_exponent, exponent):
  """cirq.PhasedXPowGate = Z**p X**t Z**-p (applied right to left)."""
  return [Z(q)**(-_as_exponent(phase_exponent)), X(q)**exponent, Z(q)**phase_exponent]


def fsim(q0, q1, theta, phi):
  """cirq.FSimGate(theta, phi) = ISWAP**(-2 theta/pi) . CZ**(-phi/pi)."""
  return [ISWAP(q0, q1)**(_as_exponent(theta) * (-2.0 / math.pi)),
          CZ(q0, q1)**(_as_exponent(phi) * (-1.0 / math.pi))]


def phased_iswap_pow(q0, q1, phase_exponent, exponent):
  """cirq.PhasedISwapPowGate = (Z**p x Z**-p) ISWAP**t (Z**-p x Z**p)."""
  p = _as_exponent(phase_exponent)
  return [Z(q0)**(-p), Z(q1)**p, ISWAP(q0, q1)**exponent, Z(q0)**p, Z(q1)**(-p)]


# ---------------------------------------------------------------------------
# Circuits
# ---------------------------------------------------------------------------
def _flatten(items) -> Iterable[Gate]:
  for it in items:
    if isinstance(it, Gate):
      yield it
    elif isinstance(it, Circuit):
      yield from it.gates
    else:
      yield from _flatten(it)


class Circuit:
  """Ordered gate list (stand-in for cirq.Circuit)."""

  def __init__(self, *items):
    self.gates: List[Gate] = list(_flatten(items))

  def __iadd__(self, other):
    self.gates.extend(_flatten([other]))
    return self

  def __add__(self, other):
    return Circuit(self.gates, other)

  def __pow__(self, power):
    if power != -1:
      raise ValueError("Only the inverse (exponent == -1) is supported.")
    return Circuit([g.inverse() for g in reversed(self.gates)])

  def all_qubits(self):
    return set(q for g in self.gates for q in g.qubits)

  def symbols(self):
    """`tfq.util.get_circuit_symbols` (circuit.py:61,201)."""
    return set(g.exponent.symbol for g in self.gates if g.exponent.symbol is not None)

  def __len__(self):
    return len(self.gates)

  def __eq__(self, other):
    return isinstance(other, Circuit) and self.gates == other.gates

  def __repr__(self):
    return "Circuit(" + ", ".join(map(repr, self.gates)) + ")"

  def flat_gates(self, qubits: Sequence[GridQubit], symbol_names: Sequence[str]):
    """Lowers to the C ABI's gate list: (kind, q0, q1, param_idx, scalar, offset[, global_shift]) --
    the seventh entry only for gates that have a global shift (rx / ry / rz).
    `qubits` fixes the qubit index (sorted qubit j = engine qubit j);
    `symbol_names[i]` is the symbol whose value sits at params[i]."""
    # memoised per (gate list, qubit order, symbol order): a training step lowers the same circuit again and again
    # (0.4 ms at config 2's size, more than the engine's whole step).  The key holds the gate objects themselves
    # (frozen dataclasses, compared by identity first), so any edit of `self.gates` misses.
    memo = self.__dict__.get("_flat_memo")
    if (memo is not None and memo[0] == tuple(self.gates)   # (element-wise `is` before `==`: microseconds when unchanged)
        and memo[1] == list(qubits) and memo[2] == list(symbol_names)):
      return memo[3]
    out = self._lower(qubits, symbol_names)
    self.__dict__["_flat_memo"] = (tuple(self.gates), list(qubits), list(symbol_names), out)
    return out

  def _lower(self, qubits, symbol_names):
    qindex = {q: i for i, q in enumerate(qubits)}
    pindex = {s: i for i, s in enumerate(symbol_names)}
    out = []
    for g in self.gates:
      e = g.exponent
      if e.symbol is not None and e.symbol not in pindex:
        raise KeyError(f"symbol {e.symbol!r} has no value")
      q0 = qindex[g.qubits[0]]
      q1 = qindex[g.qubits[1]] if len(g.qubits) > 1 else -1
      out.append((g.kind, q0, q1, pindex[e.symbol] if e.symbol is not None else -1,
                  e.scalar if e.symbol is not None else 0.0, e.offset) +
                 ((float(g.global_shift),) if g.global_shift else ()))
    return out


# ---------------------------------------------------------------------------
# Pauli sums
# ---------------------------------------------------------------------------
class PauliString:
  """coefficient * product of single-qubit Paulis (stand-in for cirq.PauliString)."""

  def __init__(self, *factors, coefficient: float = 1.0):
    self.coefficient = float(coefficient)
    self.paulis: Dict[GridQubit, str] = {}
    for f in _flatten_paulis(factors):
      if isinstance(f, PauliString):
        self.coefficient *= f.coefficient
        for q, p in f.paulis.items():
          self._mul_in(q, p)
      else:
        q, p = f
        self._mul_in(q, p)

  def _mul_in(self, q, p):
    if q in self.paulis:
      raise ValueError("repeated qubit in a PauliString is not supported")
    self.paulis[q] = p

  def __mul__(self, other):
    if isinstance(other, (int, float)):
      return PauliString(self, coefficient=other)
    return PauliString(self, other)

  __rmul__ = __mul__

  def __neg__(self):
    return self * -1.0

  def __add__(self, other):
    return PauliSum.from_pauli_strings([self]) + other

  def __sub__(self, other):
    return PauliSum.from_pauli_strings([self]) - other


def _flatten_paulis(items):
  for it in items:
    if isinstance(it, (PauliString, tuple)):
      yield it
    else:
      yield from _flatten_paulis(it)


def PX(q):  # pylint: disable=invalid-name
  return PauliString((q, "X"))


def PY(q):  # pylint: disable=invalid-name
  return PauliString((q, "Y"))


def PZ(q):  # pylint: disable=invalid-name
  return PauliString((q, "Z"))


class PauliSum:
  """Sum of PauliStrings (stand-in for cirq.PauliSum)."""

  def __init__(self, terms: Optional[List[PauliString]] = None):
    self.terms: List[PauliString] = list(terms or [])

  @staticmethod
  def from_pauli_strings(strings):
    if isinstance(strings, PauliString):
      strings = [strings]
    return PauliSum(list(strings))

  def __add__(self, other):
    if isinstance(other, PauliString):
      return PauliSum(self.terms + [other])
    return PauliSum(self.terms + other.terms)

  def __iadd__(self, other):
    self.terms = (self + other).terms
    return self

  def __sub__(self, other):
    if isinstance(other, PauliString):
      return PauliSum(self.terms + [-other])
    return PauliSum(self.terms + [-t for t in other.terms])

  def __isub__(self, other):
    self.terms = (self - other).terms
    return self

  def __mul__(self, c):
    return PauliSum([t * c for t in self.terms])

  __rmul__ = __mul__

  def qubits(self):
    return set(q for t in self.terms for q in t.paulis)

  def masks(self, qubits: Sequence[GridQubit]):
    """[(coeff, x_mask, z_mask)] in the C ABI's qubit-space convention.  Memoised per (term objects, qubit order): the
    SAME list object comes back while neither changed (the engine cache of `AnalyticQuantumInference` recognises an
    unchanged operator by identity before it falls back to comparing contents) -- treat it as read-only."""
    memo = self.__dict__.get("_masks_memo")
    if (memo is not None and memo[0] == tuple(map(id, self.terms)) and memo[1] == [t.coefficient for t in self.terms]
        and memo[2] == list(qubits)):
      return memo[3]
    qindex = {q: i for i, q in enumerate(qubits)}
    out = []
    for t in self.terms:
      x = z = 0
      for q, p in t.paulis.items():
        if p in ("X", "Y"):
          x |= 1 << qindex[q]
        if p in ("Z", "Y"):
          z |= 1 << qindex[q]
      out.append((t.coefficient, x, z))
    self.__dict__["_masks_memo"] = (tuple(map(id, self.terms)), [t.coefficient for t in self.terms], list(qubits), out)
    self.__dict__["_masks_memo_terms"] = list(self.terms)  # (keeps the ids above alive)
    return out


PauliSumLike = Union[PauliSum, PauliString]


def as_pauli_sum(op: PauliSumLike) -> PauliSum:
  if isinstance(op, PauliSum):
    return op
  # (one wrapper per string, kept with it: the wrapper carries the memo of `masks`)
  wrapped = op.__dict__.get("_as_sum")
  if wrapped is None or len(wrapped.terms) != 1 or wrapped.terms[0] is not op:
    wrapped = PauliSum.from_pauli_strings(op)
    op.__dict__["_as_sum"] = wrapped
  return wrapped


# ---------------------------------------------------------------------------
# exp(-i * coefficient * PauliSum) as a circuit (restatement of tfq.util.exponential, TFQ 0.6.1,
# used by qhbmlib/models/circuit.py:271-272 to build the QAIA ansatz)
# ---------------------------------------------------------------------------
def _strings_commute(a: PauliString, b: PauliString) -> bool:
  anti = sum(1 for q, p in a.paulis.items() if q in b.paulis and b.paulis[q] != p)
  return anti % 2 == 0


def _exponential_of_string(theta, string: PauliString) -> List[Gate]:
  """exp(-i theta c P): Clifford basis change to Z...Z (X: H, Y: rx(pi/2)), CNOT ladder onto the
  last qubit, rz(2 theta c) there, and everything undone.  rx / rz carry cirq's global_shift = -0.5,
  so the circuit IS exp(-i theta c P), not just up to a phase."""
  qubits = sorted(string.paulis)
  if not qubits:
    return []  # identity string: a global phase
  to_z = []
  for q in qubits:
    p = string.paulis[q]
    if p == "X":
      to_z.append(H(q))
    elif p == "Y":
      to_z.append(rx(math.pi / 2)(q))
  focal = qubits[-1]
  ladder = [CNOT(q, focal) for q in qubits[:-1]]
  core = rz(_as_exponent(theta) * (2.0 * string.coefficient))(focal)
  undo = [g.inverse() for g in reversed(to_z + ladder)]
  return to_z + ladder + [core] + undo


def exponential(operators, coefficients=None) -> Circuit:
  """Circuit of prod_k exp(-i coefficients[k] operators[k]), operators applied in list order.

  `operators`: PauliStrings or PauliSums whose terms commute with one another (TFQ raises
  otherwise, so does this); `coefficients`: floats, symbol names or `Symbol`s (default 1.0).
  Every emitted gate has exponent scalar*symbol + offset, the form the engine's C ABI takes.
  """
  operators = list(operators)
  if coefficients is None:
    coefficients = [1.0] * len(operators)
  coefficients = list(coefficients)
  if len(coefficients) != len(operators):
    raise ValueError("the number of coefficients must match the number of operators")
  circuit = Circuit()
  for coeff, op in zip(coefficients, operators):
    if isinstance(coeff, str):
      coeff = Symbol(coeff)
    if not isinstance(coeff, (int, float, Symbol, Exponent)):
      raise TypeError("a coefficient must be a real number, a symbol name or a Symbol")
    if isinstance(op, PauliString):
      terms = [op]
    elif isinstance(op, PauliSum):
      terms = op.terms
      for i, a in enumerate(terms):
        for b in terms[i + 1:]:
          if not _strings_commute(a, b):
            raise ValueError("the terms of a PauliSum to exponentiate must commute with one another")
    else:
      raise TypeError("an operator must be a PauliString or a PauliSum")
    for term in terms:
      circuit += _exponential_of_string(coeff, term)
  return circuit
