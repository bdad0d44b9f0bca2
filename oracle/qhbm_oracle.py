"""CPU oracle (numpy, complex128) for the QuantumInference.expectation hot path.

THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
The product path (`qhbm-library_amd/`) never imports, links or executes
anything in `oracle/`.

What it restates
----------------
The reference delegates every statevector flop to `tensorflow-quantum==0.6.1`
(C++ ops over qsim, fp32) with gate matrices defined by `cirq-core==0.14.1`
(`/root/reference/pyproject.toml:30-32`, `poetry.lock:122-123,1230-1244`).
Neither package is vendored in `/root/reference` nor importable here, so this
file restates the *published* algorithm (dense statevector simulation of cirq
power gates, Pauli-sum expectation, adjoint differentiation) and follows the
reference's own Python for everything around it:

  qhbmlib/inference/qnn.py:50-80,114-139   flow of expectation()/_expectation()
  qhbmlib/models/circuit.py:54-63,129-136  bit injection (X**bit per qubit)
  qhbmlib/models/circuit.py:138-178        append / inverse
  qhbmlib/models/circuit_utils.py:23-29    bit_circuit
  qhbmlib/models/energy.py:115-120,165-167,200-209  operator shards / expectation
  qhbmlib/models/energy_utils.py:39-110    spins, VariableDot, Parity
  qhbmlib/utils.py:43-92                   weighted_average, unique, expand
  qhbmlib/inference/ebm.py:262-329         score-function gradient
  qhbmlib/inference/vqt_loss.py:46-55, qmhl_loss.py:33-34
  tests/test_util.py:25-67                 hardware-efficient ansatz
  baselines/train.py:46-58                 TFIM ring

Parity pin
----------
Pinned by the closed-form known-answer tests the reference's own test-suite
holds for this path (tests/test_oracle_kat.py lists them with file:line):
qnn_test.py:83-180, vqt_loss_test.py:133-205, qmhl_loss_test.py:136-272,
energy_test.py:113-145,233-249, ebm_test.py:515-559, utils_test.py:47-186,
qhbm_utils_test.py:28-51.  Tests whose expected side is cirq/TFQ itself cannot
be re-run here: for arbitrary random circuits parity vs TFQ rests on the gate
definitions below.
"""

import itertools
import math

import numpy as np

# ---------------------------------------------------------------------------
# Gate kinds -- numeric values mirror include/qhbm_engine.h (checked by
# tests/test_abi.py).
# ---------------------------------------------------------------------------
GATE_I = 0
GATE_XPOW = 1
GATE_YPOW = 2
GATE_ZPOW = 3
GATE_HPOW = 4
GATE_CZPOW = 5
GATE_CNOTPOW = 6
GATE_SWAPPOW = 7
GATE_ISWAPPOW = 8
GATE_XXPOW = 9
GATE_YYPOW = 10
GATE_ZZPOW = 11

_I2 = np.eye(2, dtype=np.complex128)
_X = np.array([[0, 1], [1, 0]], dtype=np.complex128)
_Y = np.array([[0, -1j], [1j, 0]], dtype=np.complex128)
_Z = np.array([[1, 0], [0, -1]], dtype=np.complex128)
_H = (_X + _Z) / math.sqrt(2.0)
_P1 = np.array([[0, 0], [0, 1]], dtype=np.complex128)
_P0 = np.array([[1, 0], [0, 0]], dtype=np.complex128)


def _pm_components(base):
  """Eigen-components of a Hermitian involution: (0,(I+G)/2), (1,(I-G)/2).

  cirq EigenGate convention: G**t = sum_k exp(i pi t e_k) P_k, with e = 0 for
  eigenvalue +1 and e = 1 for eigenvalue -1 (cirq 0.14.1
  `cirq/ops/common_gates.py`, `_eigen_components` of XPowGate & friends).
  """
  eye = np.eye(base.shape[0], dtype=np.complex128)
  return [(0.0, (eye + base) / 2.0), (1.0, (eye - base) / 2.0)]


def _eigen_components(kind):
  """(exponent, projector) list of the base gate; first qubit = high bit."""
  if kind == GATE_I:
    return [(0.0, _I2)]
  if kind == GATE_XPOW:
    return _pm_components(_X)
  if kind == GATE_YPOW:
    return _pm_components(_Y)
  if kind == GATE_ZPOW:
    return _pm_components(_Z)
  if kind == GATE_HPOW:
    return _pm_components(_H)
  if kind == GATE_CZPOW:
    return _pm_components(np.diag([1, 1, 1, -1]).astype(np.complex128))
  if kind == GATE_CNOTPOW:
    return _pm_components(np.kron(_P0, _I2) + np.kron(_P1, _X))
  if kind == GATE_SWAPPOW:
    swap = np.array(
        [[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]],
        dtype=np.complex128)
    return _pm_components(swap)
  if kind == GATE_ISWAPPOW:
    # cirq ISwapPowGate._eigen_components: exponents 0, +1/2, -1/2.
    p0 = np.diag([1, 0, 0, 1]).astype(np.complex128)
    pp = np.zeros((4, 4), dtype=np.complex128)
    pp[1:3, 1:3] = [[0.5, 0.5], [0.5, 0.5]]
    pm = np.zeros((4, 4), dtype=np.complex128)
    pm[1:3, 1:3] = [[0.5, -0.5], [-0.5, 0.5]]
    return [(0.0, p0), (0.5, pp), (-0.5, pm)]
  if kind == GATE_XXPOW:
    return _pm_components(np.kron(_X, _X))
  if kind == GATE_YYPOW:
    return _pm_components(np.kron(_Y, _Y))
  if kind == GATE_ZZPOW:
    return _pm_components(np.kron(_Z, _Z))
  raise ValueError(f"unknown gate kind {kind}")


def gate_num_qubits(kind):
  return 1 if kind in (GATE_I, GATE_XPOW, GATE_YPOW, GATE_ZPOW,
                       GATE_HPOW) else 2


def gate_matrix(kind, t, global_shift=0.0):
  """cirq matrix of G**t (complex128)."""
  out = 0
  for e, proj in _eigen_components(kind):
    out = out + np.exp(1j * math.pi * t * (e + global_shift)) * proj
  return out


def gate_matrix_derivative(kind, t, global_shift=0.0):
  """d/dt of gate_matrix."""
  out = 0
  for e, proj in _eigen_components(kind):
    w = e + global_shift
    out = out + (1j * math.pi * w) * np.exp(1j * math.pi * t * w) * proj
  return out


# ---------------------------------------------------------------------------
# Flat circuits: list of (kind, q0, q1, param_idx, scalar, offset[, global_shift]).
# global_shift is cirq's EigenGate shift (rx/ry/rz: -0.5): the gate is
# exp(i pi t global_shift) * G**t.  It never changes an expectation value; the
# state and the unitary carry it (qnn_utils.py:23-33).
# ---------------------------------------------------------------------------
def gate_global_shift(gate):
  return float(gate[6]) if len(gate) > 6 else 0.0


def gate_exponent(gate, params):
  pidx, scalar, offset = gate[3], gate[4], gate[5]
  if pidx < 0:
    return float(offset)
  return float(scalar) * float(params[pidx]) + float(offset)


def inverse_gates(gates):
  """circuit.py:164-176: reversed order, exponent negated, same variables."""
  return [(g[0], g[1], g[2], g[3], -g[4], -g[5]) + tuple(g[6:]) for g in reversed(gates)]


def _apply_matrix(state, mat, qubits):
  """state has shape (2,)*n with axis q == qubit q (big-endian)."""
  k = len(qubits)
  m = mat.reshape((2,) * (2 * k))
  moved = np.tensordot(m, state, axes=(list(range(k, 2 * k)), list(qubits)))
  # tensordot puts the k output axes first; move them back to `qubits`.
  return np.moveaxis(moved, list(range(k)), list(qubits))


def basis_state(n, bits):
  """circuit.py:129-136 + circuit_utils.py:23-29: prod_q X(q)**bit_q |0..0>.

  X**1 == X and X**0 == I exactly, so the injector is a basis state.
  """
  state = np.zeros((2,) * n, dtype=np.complex128)
  state[tuple(int(b) for b in bits)] = 1.0
  return state


def tfq_bit_permutation(n):
  """Column j of a bitstring drives qubit perm[j] in the reference.

  circuit.py:59-62 sorts the injector's symbol *strings*
  (`bit_circuit_bit_10` < `bit_circuit_bit_2`) and circuit.py:132-134 assigns
  column j to the j-th sorted name.  Identity for n <= 10.
  """
  names = [f"bit_circuit_bit_{i}" for i in range(n)]
  return [int(s.rsplit("_", 1)[1]) for s in sorted(names)]


def apply_bit_order(bitstrings, tfq_compat):
  bitstrings = np.asarray(bitstrings)
  if not tfq_compat:
    return bitstrings
  n = bitstrings.shape[1]
  perm = tfq_bit_permutation(n)
  out = np.zeros_like(bitstrings)
  for j, q in enumerate(perm):
    out[:, q] = bitstrings[:, j]
  return out


def simulate(n, gates, params, bits):
  """Final statevector, shape (2,)*n, of gates applied to |bits>."""
  state = basis_state(n, bits)
  for g in gates:
    kind, q0, q1 = g[0], g[1], g[2]
    t = gate_exponent(g, params)
    qs = (q0,) if gate_num_qubits(kind) == 1 else (q0, q1)
    state = _apply_matrix(state, gate_matrix(kind, t, gate_global_shift(g)), qs)
  return state


# ---------------------------------------------------------------------------
# Pauli sums: op = list of (coeff, x_mask, z_mask) in QUBIT space
# (bit q of mask <-> qubit q), see include/qhbm_engine.h.
# ---------------------------------------------------------------------------
def pauli_term(coeff, paulis):
  """paulis: dict or iterable of (qubit, 'X'|'Y'|'Z')."""
  x = 0
  z = 0
  items = paulis.items() if isinstance(paulis, dict) else paulis
  for q, p in items:
    if p in ("X", "Y"):
      x |= 1 << q
    if p in ("Z", "Y"):
      z |= 1 << q
  return (float(coeff), x, z)


def apply_pauli(state, x_mask, z_mask):
  n = state.ndim
  out = state
  for q in range(n):
    xb = (x_mask >> q) & 1
    zb = (z_mask >> q) & 1
    if xb and zb:
      out = _apply_matrix(out, _Y, (q,))
    elif xb:
      out = _apply_matrix(out, _X, (q,))
    elif zb:
      out = _apply_matrix(out, _Z, (q,))
  return out


def apply_op(state, op):
  out = np.zeros_like(state)
  for coeff, x, z in op:
    out = out + coeff * apply_pauli(state, x, z)
  return out


def op_expectation(state, op):
  return float(np.real(np.vdot(state.ravel(), apply_op(state, op).ravel())))


# ---------------------------------------------------------------------------
# The hot path: qnn.py:50-80 + tfq.layers.Expectation (qnn.py:134-138).
# ---------------------------------------------------------------------------
def unique_bitstrings_with_counts(bitstrings):
  """utils.py:61-78 (tf.raw_ops.UniqueWithCountsV2, axis 0): unique rows in
  FIRST-OCCURRENCE order (pinned by tests/utils_test.py:165-167)."""
  bitstrings = np.asarray(bitstrings)
  seen = {}
  uniq = []
  idx = np.zeros(bitstrings.shape[0], dtype=np.int32)
  counts = []
  for i, row in enumerate(bitstrings):
    key = row.tobytes()
    j = seen.get(key)
    if j is None:
      j = len(uniq)
      seen[key] = j
      uniq.append(row)
      counts.append(0)
    idx[i] = j
    counts[j] += 1
  y = (np.stack(uniq) if uniq else np.zeros((0,) + bitstrings.shape[1:],
                                             bitstrings.dtype))
  return y, idx, np.asarray(counts, dtype=np.int32)


def expand_unique_results(y, idx):
  """utils.py:81-92."""
  return np.asarray(y)[np.asarray(idx)]


def weighted_average(counts, values):
  """utils.py:43-58."""
  counts = np.asarray(counts, dtype=np.float64)
  values = np.asarray(values, dtype=np.float64)
  return np.tensordot(counts, values, axes=(0, 0)) / counts.sum()


def expectation(n, gates, params, bitstrings, ops, tfq_compat_bit_order=False):
  """[B, n_ops] of <x|C^dag O_k C|x>; rows in input order (qnn.py:50-80)."""
  bitstrings = apply_bit_order(np.asarray(bitstrings), tfq_compat_bit_order)
  uniq, idx, _ = unique_bitstrings_with_counts(bitstrings)
  vals = np.zeros((uniq.shape[0], len(ops)))
  for u, bits in enumerate(uniq):
    psi = simulate(n, gates, params, bits)
    for k, op in enumerate(ops):
      vals[u, k] = op_expectation(psi, op)
  return expand_unique_results(vals, idx)


def expectation_jacobian(n, gates, params, bitstrings, ops,
                         tfq_compat_bit_order=False):
  """Values [B,T] and exact Jacobian [B,T,P] by adjoint differentiation.

  For psi_g = U_g psi_{g-1} and lam_g = U_{g+1}^dag ... U_G^dag O psi_G:
      dE/dt_g = 2 Re <lam_g| dU_g/dt |psi_{g-1}>
  and dE/dparams[p] = sum_{g: param_idx_g = p} scalar_g dE/dt_g, which is what
  TF obtains by summing symbol-value gradients through the tile of qnn.py:75-76.
  """
  bitstrings = apply_bit_order(np.asarray(bitstrings), tfq_compat_bit_order)
  n_params = len(params)
  vals = np.zeros((bitstrings.shape[0], len(ops)))
  jac = np.zeros((bitstrings.shape[0], len(ops), n_params))
  for b, bits in enumerate(bitstrings):
    psi_final = simulate(n, gates, params, bits)
    for k, op in enumerate(ops):
      lam = apply_op(psi_final, op)
      vals[b, k] = float(np.real(np.vdot(psi_final.ravel(), lam.ravel())))
      psi = psi_final
      for g in reversed(gates):
        kind, q0, q1, pidx, scalar = g[:5]
        shift = gate_global_shift(g)
        t = gate_exponent(g, params)
        qs = (q0,) if gate_num_qubits(kind) == 1 else (q0, q1)
        u_dag = gate_matrix(kind, t, shift).conj().T
        psi = _apply_matrix(psi, u_dag, qs)  # psi_{g-1}
        if pidx >= 0:
          dpsi = _apply_matrix(psi, gate_matrix_derivative(kind, t, shift), qs)
          jac[b, k, pidx] += scalar * 2.0 * float(
              np.real(np.vdot(lam.ravel(), dpsi.ravel())))
        lam = _apply_matrix(lam, u_dag, qs)
  return vals, jac


def expectation_parameter_shift(n, gates, params, bitstrings, ops):
  """Jacobian by the two-term shift rule for exponent c*s:
  dE/ds = (pi c / 2) [E(s + 1/(2c)) - E(s - 1/(2c))]  per gate occurrence
  (baselines/train.py:190-240 with shift 0.5, scale pi/2; tfq ParameterShift).
  Valid for gates whose eigen-exponents differ by 1 (all kinds but ISWAPPOW).
  """
  bitstrings = np.asarray(bitstrings)
  jac = np.zeros((bitstrings.shape[0], len(ops), len(params)))
  for gi, g in enumerate(gates):
    kind, q0, q1, pidx, scalar, offset = g[:6]
    if pidx < 0:
      continue
    if kind == GATE_ISWAPPOW:
      raise ValueError("two-term shift rule does not apply to ISWAPPOW")
    for sign in (+1.0, -1.0):
      shifted = list(gates)
      shifted[gi] = (kind, q0, q1, pidx, scalar, offset + sign * 0.5) + tuple(g[6:])
      e = expectation(n, shifted, params, bitstrings, ops)
      jac[:, :, pidx] += sign * (math.pi * scalar / 2.0) * e
  return jac


# ---------------------------------------------------------------------------
# Model builders named by BASELINE.json configs.
# ---------------------------------------------------------------------------
def hea_symbol_names(n, num_layers, name):
  """Symbol names in circuit order (tests/test_util.py:35-67)."""
  names = []
  for layer in range(num_layers):
    for q in range(n):
      names += [f"sx_{name}_{layer}_{q}", f"sz_{name}_{layer}_{q}"]
    if n > 1:
      for k in range(len(range(0, n - 1, 2))):
        names.append(f"sc_{name}_{layer}_{2 * k}")
      for k in range(len(range(1, n - 1, 2))):
        names.append(f"sc_{name}_{layer}_{2 * k + 1}")
  return names


def hea_gates(n, num_layers, name="m"):
  """Hardware-efficient ansatz of tests/test_util.py:25-67 as a flat circuit.

  Per layer: X**sx, Z**sz on every qubit, then CZ**sc on pairs (0,1),(2,3),..
  followed by (1,2),(3,4),...  Parameters are laid out as
  `DirectQuantumCircuit` does: index = rank of the symbol name in
  sorted(names) (circuit.py:201-204).
  Returns (gates, sorted_symbol_names).
  """
  names = hea_symbol_names(n, num_layers, name)
  order = {s: i for i, s in enumerate(sorted(names))}
  gates = []
  for layer in range(num_layers):
    for q in range(n):
      gates.append((GATE_XPOW, q, -1, order[f"sx_{name}_{layer}_{q}"], 1.0, 0.0))
      gates.append((GATE_ZPOW, q, -1, order[f"sz_{name}_{layer}_{q}"], 1.0, 0.0))
    if n > 1:
      for k, q0 in enumerate(range(0, n - 1, 2)):
        gates.append((GATE_CZPOW, q0, q0 + 1,
                      order[f"sc_{name}_{layer}_{2 * k}"], 1.0, 0.0))
      for k, q0 in enumerate(range(1, n - 1, 2)):
        gates.append((GATE_CZPOW, q0, q0 + 1,
                      order[f"sc_{name}_{layer}_{2 * k + 1}"], 1.0, 0.0))
  return gates, sorted(names)


def tfim_ring_op(n, bias=1.0):
  """baselines/train.py:52-58 (1-D): H = -bias sum X_i - sum Z_i Z_{i+1}, periodic."""
  terms = []
  for i in range(n):
    terms.append(pauli_term(-bias, [(i, "X")]))
  for i in range(n):
    j = (i + 1) % n
    if i == j:
      continue
    terms.append(pauli_term(-1.0, [(i, "Z"), (j, "Z")]))
  return terms


def xxz_chain_op(n, delta=0.5):
  """Open XXZ chain sum_i (X_i X_{i+1} + Y_i Y_{i+1} + delta Z_i Z_{i+1}).

  Not defined in the reference; BASELINE.json config 3 names it and
  SURVEY.md section 8(d) fixes this definition."""
  terms = []
  for i in range(n - 1):
    terms.append(pauli_term(1.0, [(i, "X"), (i + 1, "X")]))
    terms.append(pauli_term(1.0, [(i, "Y"), (i + 1, "Y")]))
    terms.append(pauli_term(delta, [(i, "Z"), (i + 1, "Z")]))
  return terms


def random_pauli_op(n, num_terms, seed, p_identity=0.75):
  """BASELINE.json config 4: each qubit in {I,X,Y,Z} with
  P(I)=p_identity, at least one non-identity, coefficients N(0,1)."""
  rng = np.random.default_rng(seed)
  terms = []
  while len(terms) < num_terms:
    paulis = []
    for q in range(n):
      if rng.random() >= p_identity:
        paulis.append((q, "XYZ"[rng.integers(3)]))
    if not paulis:
      continue
    terms.append(pauli_term(rng.normal(), paulis))
  return terms


# ---------------------------------------------------------------------------
# Energy functions (classical side) -- energy.py / energy_utils.py.
# ---------------------------------------------------------------------------
def spins_from_bitstrings(bits):
  """energy_utils.py:39-52: |0> -> +1, |1> -> -1."""
  return 1.0 - 2.0 * np.asarray(bits, dtype=np.float64)


def parity_indices(num_bits, order):
  """energy_utils.py:97-102: all i-subsets, i = 1..order, combinations order."""
  out = []
  for i in range(1, order + 1):
    out.extend(itertools.combinations(range(num_bits), i))
  return out


def parities(bits, indices):
  """energy_utils.py:104-110."""
  s = spins_from_bitstrings(bits)
  return np.stack([np.prod(s[..., list(ix)], axis=-1) for ix in indices], -1)


def bernoulli_energy(bits, thetas):
  """energy.py:141-144: sum_i theta_i (1 - 2 b_i)."""
  return spins_from_bitstrings(bits) @ np.asarray(thetas, dtype=np.float64)


def kobe_energy(bits, thetas, order):
  bits = np.asarray(bits)
  ix = parity_indices(bits.shape[-1], order)
  return parities(bits, ix) @ np.asarray(thetas, dtype=np.float64)


def bernoulli_shards(n):
  """energy.py:165-167: Z_q for every qubit."""
  return [[pauli_term(1.0, [(q, "Z")])] for q in range(n)]


def kobe_shards(n, order):
  """energy.py:200-209."""
  return [[pauli_term(1.0, [(q, "Z") for q in ix])]
          for ix in parity_indices(n, order)]


def all_bitstrings(n):
  """ebm.py:445-447: itertools.product([0,1], repeat=n) (big-endian)."""
  return np.array(list(itertools.product([0, 1], repeat=n)), dtype=np.int8)


def log_partition_exact(energy_fn, n):
  e = energy_fn(all_bitstrings(n))
  m = (-e).max()
  return float(m + np.log(np.exp(-e - m).sum()))


def entropy_exact(energy_fn, n):
  e = energy_fn(all_bitstrings(n))
  logits = -e - (-e).max()
  p = np.exp(logits)
  p /= p.sum()
  nz = p > 0
  return float(-(p[nz] * np.log(p[nz])).sum())


def modular_hamiltonian_expectation(n, circuit_gates, ham_gates, params,
                                    bitstrings, shards, thetas):
  """qnn.py:69-72,120-127: circuit + hamiltonian.circuit_dagger, measure the
  Z-string shards, combine with VariableDot (energy_utils.py:79-81).
  Returns [B, 1]."""
  total = list(circuit_gates) + inverse_gates(ham_gates)
  shard_vals = expectation(n, total, params, bitstrings, shards)
  return (shard_vals @ np.asarray(thetas, dtype=np.float64))[:, None]


# ---------------------------------------------------------------------------
# Losses for a GIVEN multiset of EBM samples (the sampler is outside the path).
# ---------------------------------------------------------------------------
def vqt_loss_and_grads(n, gates, params, samples, target_op, beta, energy_fn,
                       energy_grad_fn, log_partition):
  """vqt_loss.py:46-55 with ebm.py:262-329 evaluated on fixed `samples`.

  energy_grad_fn(bits) -> [U, n_theta] Jacobian of the energy.
  Returns (loss, dloss/dtheta, dloss/dparams)."""
  uniq, _, counts = unique_bitstrings_with_counts(samples)
  h, jac = expectation_jacobian(n, gates, params, uniq, [target_op])
  h = h[:, 0]
  f = beta * h - energy_fn(uniq)  # energies are stop_gradient
  avg_f = weighted_average(counts, f)
  loss = avg_f - log_partition  # log partition is stop_gradient
  # ebm.py:303-324 with upstream = 1: poa - aop + function_grads.
  e_grads = energy_grad_fn(uniq)
  avg_e_grads = weighted_average(counts, e_grads)
  avg_prod = weighted_average(counts, e_grads * f[:, None])
  dtheta = avg_e_grads * avg_f - avg_prod
  dparams = beta * weighted_average(counts, jac[:, 0, :])
  return float(loss), dtheta, dparams


# ---------------------------------------------------------------------------
# Dense metrics: qnn_utils.py:23-33 unitary, ebm_utils.py:24-36
# probabilities, qhbm_utils.py:24-116 density_matrix / fidelity.
# ---------------------------------------------------------------------------
def unitary(n, gates, params):
  """Matrix of the circuit: column x is simulate(|x>), indices big-endian."""
  cols = [simulate(n, gates, params, list(b)).ravel() for b in itertools.product([0, 1], repeat=n)]
  return np.stack(cols, axis=1)


def probabilities(energy_fn, n):
  """exp(-E(x)) / Z over all bitstrings in itertools.product order (ebm_utils.py:24-36)."""
  e = np.exp(-energy_fn(all_bitstrings(n)))
  return e / e.sum()


def density_matrix(n, gates, params, energy_fn):
  """rho = U diag(p) U^dagger (qhbm_utils.py:57-59)."""
  u = unitary(n, gates, params)
  return (u * probabilities(energy_fn, n)[None, :]) @ u.conj().T


def _sqrtm_psd(m):
  w, v = np.linalg.eigh((m + m.conj().T) / 2)
  return (v * np.sqrt(np.clip(w, 0, None))[None, :]) @ v.conj().T


def fidelity_direct(rho, sigma):
  """(tr sqrt(sqrt(rho) sigma sqrt(rho)))^2, the direct formula the reference test compares
  with (tests/inference/qhbm_utils_test.py:83-88)."""
  s = _sqrtm_psd(rho)
  return float(np.real(np.trace(_sqrtm_psd(s @ sigma @ s)))**2)
