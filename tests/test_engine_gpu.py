"""Parity of the HIP engine (through the C ABI) against the numpy oracle.

GPU only.  Tolerances: the engine computes in fp32 (complex64 statevectors);
the oracle in complex128.  Bars (SURVEY.md 8c / BASELINE.md):
  expectations  |d| <= 1e-5 * sum|c_k|  (n <= 12),  5e-5 * sum|c_k| deeper/larger
  gradients     |d| <= 1e-4 * max(1, ||grad||_inf)
"""
import itertools
import math

import numpy as np
import pytest
import torch

from oracle import qhbm_oracle as O
from qhbmlib_amd import _engine as E

pytestmark = pytest.mark.gpu


def _op_norm(ops):
  return np.array([sum(abs(c) for c, _, _ in op) for op in ops])


def _engine(n, gates, n_params, ops, **options):
  eng = E.Engine(0)
  for k, v in options.items():
    eng.set_option(k, v)
  eng.set_circuit(n, gates, n_params)
  eng.set_observables(ops)
  return eng


def _random_bits(rng, count, n):
  return rng.integers(0, 2, size=(count, n)).astype(np.int8)


def random_circuit(rng, n, n_gates, n_params, kinds=None):
  """Random flat circuit over every gate kind (the TFQ-serialisable set)."""
  kinds = kinds or list(range(12))
  gates = []
  for _ in range(n_gates):
    kind = int(rng.choice(kinds))
    q0 = int(rng.integers(n))
    q1 = -1
    if O.gate_num_qubits(kind) == 2:
      q1 = int(rng.integers(n - 1))
      if q1 >= q0:
        q1 += 1
    if rng.random() < 0.8:
      gates.append((kind, q0, q1, int(rng.integers(n_params)),
                    float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-0.5, 0.5))))
    else:
      gates.append((kind, q0, q1, -1, 0.0, float(rng.uniform(-1, 1))))
  return gates


def check_values(eng, n, gates, params, bits, ops, rel=1e-5):
  got = eng.expectation(bits, params).cpu().numpy()
  want = O.expectation(n, gates, params, bits, ops)
  tol = rel * np.maximum(_op_norm(ops), 1.0)
  assert got.shape == want.shape
  err = np.abs(got - want)
  assert (err <= tol[None, :]).all(), f"max err {err.max()} tol {tol}"
  return got, want


def check_jacobian(eng, n, gates, params, bits, ops, rel=1e-4):
  vals, jac = eng.expectation_jacobian(bits, params)
  want_vals, want_jac = O.expectation_jacobian(n, gates, params, bits, ops)
  scale = max(1.0, np.abs(want_jac).max())
  np.testing.assert_allclose(jac.cpu().numpy(), want_jac, atol=rel * scale, rtol=0)
  np.testing.assert_allclose(vals.cpu().numpy(), want_vals,
                             atol=1e-5 * max(1.0, _op_norm(ops).max()), rtol=0)
  return want_jac


# ---- BASELINE.json config 1: 4-qubit TFIM, depth-2 HEA, 32 samples -----------
def test_c1_tfim_hea_all_paths():
  n, layers = 4, 2
  rng = np.random.default_rng(0)
  gates, names = O.hea_gates(n, layers, "c1")
  params = rng.uniform(-1, 1, len(names))
  ops = [O.tfim_ring_op(n)]
  bits = _random_bits(rng, 32, n)
  eng = _engine(n, gates, len(names), ops)
  check_values(eng, n, gates, params, bits, ops)
  want_jac = check_jacobian(eng, n, gates, params, bits, ops)
  up = rng.normal(size=(32, 1))
  want_grad = np.einsum("bt,btp->p", up, want_jac)
  for method in (E.GRAD_ADJOINT, E.GRAD_PARAMETER_SHIFT):
    vals, grad = eng.expectation_vjp(bits, params, up, method)
    scale = max(1.0, np.abs(want_grad).max())
    np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=1e-4 * scale, rtol=0)
    np.testing.assert_allclose(vals.cpu().numpy(),
                               O.expectation(n, gates, params, bits, ops), atol=1e-5 * 8)


# ---- reference KAT through the engine: X**p (qnn_test.py:83-180) --------------
def test_x_pow_kat():
  n, p = 3, 0.37
  gates = [(O.GATE_XPOW, q, -1, 0, 1.0, 0.0) for q in range(n)]
  bits = O.all_bitstrings(n)
  sin, cos = math.sin(math.pi * p), math.cos(math.pi * p)
  for pauli, val, grad in (
      ("X", lambda s: 0.0, lambda s: 0.0),
      ("Y", lambda s: -((-1.0)**s) * sin, lambda s: -((-1.0)**s) * math.pi * cos),
      ("Z", lambda s: ((-1.0)**s) * cos, lambda s: -((-1.0)**s) * math.pi * sin)):
    ops = [[O.pauli_term(1.0, [(q, pauli)])] for q in range(n)]
    eng = _engine(n, gates, 1, ops)
    vals, jac = eng.expectation_jacobian(bits, [p])
    np.testing.assert_allclose(vals.cpu().numpy(),
                               [[val(s) for s in row] for row in bits], atol=1e-6)
    np.testing.assert_allclose(jac.cpu().numpy()[:, :, 0],
                               [[grad(s) for s in row] for row in bits], atol=2e-5)


# ---- every gate kind, single tile ---------------------------------------------
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_all_gate_kinds_small(seed):
  n, n_params = 5, 6
  rng = np.random.default_rng(seed)
  gates = random_circuit(rng, n, 40, n_params)
  params = rng.uniform(-1, 1, n_params)
  ops = [O.random_pauli_op(n, 6, seed + 10, p_identity=0.4), O.tfim_ring_op(n),
         O.xxz_chain_op(n)] + O.kobe_shards(n, 2)[:4]
  bits = O.all_bitstrings(n)
  eng = _engine(n, gates, n_params, ops)
  check_values(eng, n, gates, params, bits, ops)
  check_jacobian(eng, n, gates, params, bits, ops, rel=2e-4)


# ---- multi-tile scheduling at oracle-checkable sizes ---------------------------
@pytest.mark.parametrize("n,tile,adj_tile", [(12, 10, 10), (13, 11, 10), (14, 12, 11), (15, 13, 12),
                                             (15, 10, 10), (14, 14, 13)])
def test_hea_multi_tile(n, tile, adj_tile):
  rng = np.random.default_rng(n * 100 + tile)
  gates, names = O.hea_gates(n, 3, "mt")
  params = rng.uniform(-1, 1, len(names))
  ops = [O.xxz_chain_op(n), O.tfim_ring_op(n)]
  bits = _random_bits(rng, 3, n)
  eng = _engine(n, gates, len(names), ops, tile_qubits=tile, adjoint_tile_qubits=adj_tile)
  f, b = eng.num_passes()
  if n > tile:
    assert f > 1 and b > 1
  check_values(eng, n, gates, params, bits, ops, rel=2e-5)
  up = rng.normal(size=(3, 2))
  want_vals, want_jac = O.expectation_jacobian(n, gates, params, bits, ops)
  want_grad = np.einsum("bt,btp->p", up, want_jac)
  _, grad = eng.expectation_vjp(bits, params, up)
  np.testing.assert_allclose(grad.cpu().numpy(), want_grad,
                             atol=2e-4 * max(1.0, np.abs(want_grad).max()), rtol=0)


@pytest.mark.parametrize("seed", [5, 6])
def test_all_gate_kinds_multi_tile(seed):
  n, n_params = 12, 8
  rng = np.random.default_rng(seed)
  gates = random_circuit(rng, n, 60, n_params)
  params = rng.uniform(-1, 1, n_params)
  ops = [O.random_pauli_op(n, 12, seed, p_identity=0.7), O.xxz_chain_op(n)]
  bits = _random_bits(rng, 2, n)
  eng = _engine(n, gates, n_params, ops, tile_qubits=10, adjoint_tile_qubits=10)
  check_values(eng, n, gates, params, bits, ops, rel=2e-5)
  check_jacobian(eng, n, gates, params, bits, ops, rel=3e-4)


# ---- BASELINE.json config 2 shape: 12 qubits, depth-8 HEA, TFIM ----------------
def test_c2_shape_values_and_grad():
  n, layers = 12, 8
  rng = np.random.default_rng(12)
  gates, names = O.hea_gates(n, layers, "c2")
  params = rng.uniform(-1, 1, len(names))
  ops = [O.tfim_ring_op(n)]
  bits = _random_bits(rng, 4, n)
  eng = _engine(n, gates, len(names), ops)
  check_values(eng, n, gates, params, bits, ops)
  up = rng.normal(size=(4, 1))
  _, want_jac = O.expectation_jacobian(n, gates, params, bits, ops)
  want_grad = np.einsum("bt,btp->p", up, want_jac)
  _, grad = eng.expectation_vjp(bits, params, up)
  np.testing.assert_allclose(grad.cpu().numpy(), want_grad,
                             atol=1e-4 * max(1.0, np.abs(want_grad).max()), rtol=0)


# ---- modular-Hamiltonian branch (qnn.py:69-72,120-127): U + V^dagger, Z shards --
@pytest.mark.parametrize("order", [1, 2])
def test_modular_hamiltonian_shards(order):
  n = 6
  rng = np.random.default_rng(order)
  u_gates, u_names = O.hea_gates(n, 2, "u")
  v_gates, v_names = O.hea_gates(n, 2, "v")
  p_u = len(u_names)
  v_shift = [(k, q0, q1, p + p_u, s, o) for (k, q0, q1, p, s, o) in v_gates]
  total = u_gates + O.inverse_gates(v_shift)
  params = rng.uniform(-1, 1, p_u + len(v_names))
  shards = O.bernoulli_shards(n) if order == 1 else O.kobe_shards(n, 2)
  bits = O.all_bitstrings(n)
  eng = _engine(n, total, len(params), shards)
  check_values(eng, n, total, params, bits, shards)
  check_jacobian(eng, n, total, params, bits, shards)


# ---- edge cases ------------------------------------------------------------------
def test_empty_batch_and_duplicates_and_empty_circuit():
  n = 4
  ops = [O.tfim_ring_op(n)]
  gates, names = O.hea_gates(n, 1, "e")
  params = np.linspace(-0.5, 0.5, len(names))
  eng = _engine(n, gates, len(names), ops)
  out = eng.expectation(np.zeros((0, n), np.int8), params)
  assert tuple(out.shape) == (0, 1)
  bits = np.array([[1, 0, 1, 1]] * 5 + [[0, 0, 0, 0]], np.int8)
  got = eng.expectation(bits, params).cpu().numpy()
  np.testing.assert_allclose(got, O.expectation(n, gates, params, bits, ops), atol=1e-5)
  assert np.all(got[:5] == got[0])
  # no gates at all: <x| H |x>
  eng2 = _engine(n, [], 0, ops)
  got = eng2.expectation(bits, []).cpu().numpy()
  np.testing.assert_allclose(got, O.expectation(n, [], [], bits, ops), atol=1e-6)


def test_errors_are_loud():
  eng = E.Engine(0)
  with pytest.raises(E.EngineError):
    eng.set_circuit(3, [(99, 0, -1, -1, 0.0, 0.0)], 0)
  with pytest.raises(E.EngineError):
    eng.set_circuit(3, [(O.GATE_XPOW, 5, -1, -1, 0.0, 0.0)], 0)
  eng.set_circuit(3, [(O.GATE_XPOW, 0, -1, 0, 1.0, 0.0)], 1)
  with pytest.raises(E.EngineError):
    eng.expectation(np.zeros((1, 3), np.int8), [0.1])  # no observables yet
  with pytest.raises(E.EngineError):
    eng.set_observables([[(1.0, 1 << 3, 0)]])  # qubit 3 does not exist
  with pytest.raises(ValueError):
    eng.set_observables([[(1.0, 1, 0)]])
    eng.expectation(np.zeros((1, 4), np.int8), [0.1])


def test_chunked_execution_matches():
  n = 11
  rng = np.random.default_rng(4)
  gates, names = O.hea_gates(n, 2, "ch")
  params = rng.uniform(-1, 1, len(names))
  ops = [O.xxz_chain_op(n)]
  bits = _random_bits(rng, 7, n)
  eng = _engine(n, gates, len(names), ops, tile_qubits=10, adjoint_tile_qubits=10, chunk_states=3)
  check_values(eng, n, gates, params, bits, ops, rel=2e-5)
  up = rng.normal(size=(7, 1))
  _, want_jac = O.expectation_jacobian(n, gates, params, bits, ops)
  _, grad = eng.expectation_vjp(bits, params, up)
  want = np.einsum("bt,btp->p", up, want_jac)
  np.testing.assert_allclose(grad.cpu().numpy(), want, atol=2e-4 * max(1, np.abs(want).max()), rtol=0)


def test_batch_larger_than_a_grid_dimension():
  """70 000 states of a 10-qubit circuit: more than the 65 535 a grid y-dimension holds, so the
  engine must chunk; every state is checked against the closed form of X**p."""
  n, states = 10, 70000
  gates = [(E.GATE_XPOW, q, -1, 0, 1.0, 0.0) for q in range(n)]
  ops = [[(1.0, 0, 1 << q)] for q in range(n)]
  eng = _engine(n, gates, 1, ops)
  rng = np.random.default_rng(0)
  bits = rng.integers(0, 2, size=(states, n)).astype(np.int8)
  p = np.array([0.37], np.float32)
  vals = eng.expectation(bits, p).cpu().numpy()
  want = (1.0 - 2.0 * bits) * math.cos(math.pi * 0.37)
  np.testing.assert_allclose(vals, want, atol=1e-5)
  up = np.full((states, n), 1.0 / states, np.float32)
  vals2, grad = eng.expectation_vjp(bits, p, up)
  np.testing.assert_allclose(vals2.cpu().numpy(), want, atol=1e-5)
  want_g = float(((1.0 - 2.0 * bits) * (-math.pi * math.sin(math.pi * 0.37))).sum() / states)
  np.testing.assert_allclose(grad.cpu().numpy()[0], want_g, atol=1e-4 * max(1.0, abs(want_g)))


def test_retained_forward_state():
  """qhbm_expectation_retain + qhbm_expectation_vjp_retained equal the one-call VJP, the states
  are consumed by one backward sweep, any other call drops them, and a batch that does not fit one
  backward chunk is simply not retained."""
  n, layers = 14, 3
  gates, names = O.hea_gates(n, layers, "r")
  rng = np.random.default_rng(9)
  params = rng.uniform(-1, 1, len(names)).astype(np.float32)
  ops = [O.xxz_chain_op(n), O.tfim_ring_op(n)]
  bits = _random_bits(rng, 6, n)
  up = rng.normal(size=(6, 2)).astype(np.float32)
  eng = _engine(n, gates, len(names), ops, tile_qubits=11, adjoint_tile_qubits=10)
  want_vals, want_grad = eng.expectation_vjp(bits, params, up)
  vals = eng.expectation(bits, params, retain=True)
  assert eng.retained is not None
  grad = eng.expectation_vjp_retained(bits, params, up)
  np.testing.assert_allclose(vals.cpu().numpy(), want_vals.cpu().numpy(), atol=1e-6)
  np.testing.assert_allclose(grad.cpu().numpy(), want_grad.cpu().numpy(), atol=2e-6 * max(1.0, float(want_grad.abs().max())))
  assert eng.retained is None
  with pytest.raises(E.EngineError, match="no retained forward state"):
    eng.expectation_vjp_retained(bits, params, up)            # consumed
  eng.expectation(bits, params, retain=True)
  eng.expectation(bits[:3], params)                            # another call drops the states
  with pytest.raises(E.EngineError, match="no retained forward state"):
    eng.expectation_vjp_retained(bits, params, up)
  eng.set_option("chunk_states", 4)                            # 6 states do not fit one chunk
  vals = eng.expectation(bits, params, retain=True)
  np.testing.assert_allclose(vals.cpu().numpy(), want_vals.cpu().numpy(), atol=1e-6)
  with pytest.raises(E.EngineError, match="no retained forward state"):
    eng.expectation_vjp_retained(bits, params, up)


def test_pauli_terms_wider_than_a_tile():
  """TFQ takes any PauliSum (qnn.py:134-138): a term that flips more qubits than one LDS tile
  holds is measured by the strided-gather kernel on the final state (slow path, still on the GPU);
  the adjoint forms lambda = O psi with global gathers for any mask."""
  rng = np.random.default_rng(99)
  n, layers = 14, 2
  gates, names = O.hea_gates(n, layers, "w")
  params = rng.uniform(-1, 1, len(names))
  wide = [O.pauli_term(1.0, [(q, "X") for q in range(12)])]                       # 12 flips > 9
  mixed = [O.pauli_term(0.5, [(q, "XYZ"[q % 3]) for q in range(n)]),            # 10 flips
           O.pauli_term(-0.7, [(0, "Z"), (5, "X")]), O.pauli_term(0.3, [(q, "Y") for q in range(2, 13)])]
  ops = [wide, mixed, O.tfim_ring_op(n)]
  bits = _random_bits(rng, 5, n)
  eng = _engine(n, gates, len(names), ops, tile_qubits=10, adjoint_tile_qubits=10)
  vals, jac = eng.expectation_jacobian(bits, params)
  want, want_jac = O.expectation_jacobian(n, gates, params, bits, ops)
  np.testing.assert_allclose(vals.cpu().numpy(), want, atol=1e-5 * max(1.0, _op_norm(ops).max()))
  np.testing.assert_allclose(jac.cpu().numpy(), want_jac, atol=1e-4 * max(1.0, np.abs(want_jac).max()))
  # the done-criterion of VERDICT r1 item 9: a 16-qubit all-X string at n = 20 (default tiles, K = 13)
  from oracle import qhbm_cpu as C
  n = 20
  gates, names = O.hea_gates(n, 2, "w")
  params = rng.uniform(-1, 1, len(names)).astype(np.float32)
  ops = [[O.pauli_term(1.0, [(q, "X") for q in range(2, 18)])],
         [O.pauli_term(0.8, [(q, "Y" if q % 2 else "X") for q in range(16)]), O.pauli_term(0.4, [(19, "Z")])]]
  bits = _random_bits(rng, 3, n)
  up = rng.normal(size=(3, 2)).astype(np.float32)
  eng = _engine(n, gates, len(names), ops)
  vals, grad = eng.expectation_vjp(bits, params, up)
  want, want_grad = C.expectation_vjp(n, gates, params, bits, ops, up)
  np.testing.assert_allclose(vals.cpu().numpy(), want, atol=2e-5)
  np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=1e-4 * max(1.0, np.abs(want_grad).max()))


@pytest.mark.parametrize("n,layers", [(9, 2), (16, 2), (20, 1)])
def test_masks_whose_terms_cancel_on_part_of_the_index_space(n, layers):
  """lambda = O psi skips the partner runs of a mask on the blocks (and slot pairs) where the weights of its
  terms add up to exactly zero: XX + YY on two qubits vanishes where their bits are equal.  The skip must not
  fire when the cancellation is only partial (unequal coefficients, different operators with different
  upstream weights, a Z factor on a thread bit), and a zero upstream weight kills every mask of its operator.
  Single observable (values from lambda, unweighted) and several (weighted), against the C oracle."""
  from oracle import qhbm_cpu as C
  rng = np.random.default_rng(n)
  gates, names = O.hea_gates(n, layers, "z")
  params = rng.uniform(-1, 1, len(names)).astype(np.float32)
  pairs = [(a, b) for a, b in [(0, 1), (n - 2, n - 1), (n - 1, 0), (n // 2, n - 1), (3, n - 3)] if a != b]
  flipflop = [t for a, b in pairs for t in (O.pauli_term(1.0, [(a, "X"), (b, "X")]), O.pauli_term(1.0, [(a, "Y"), (b, "Y")]))]
  unequal = [t for a, b in pairs for t in (O.pauli_term(0.75, [(a, "X"), (b, "X")]), O.pauli_term(-0.5, [(a, "Y"), (b, "Y")]))]
  dressed = [t for a, b in pairs[1:] for t in (O.pauli_term(1.0, [(a, "X"), (b, "X"), (2, "Z")]),
                                                O.pauli_term(1.0, [(a, "Y"), (b, "Y"), (2, "Z")]))]
  xx_only = [O.pauli_term(1.0, [(a, "X"), (b, "X")]) for a, b in pairs]
  yy_only = [O.pauli_term(1.0, [(a, "Y"), (b, "Y")]) for a, b in pairs]
  bits = _random_bits(rng, 6, n)
  for op in (flipflop, unequal, flipflop + dressed + O.xxz_chain_op(n)):          # values from lambda = O psi
    eng = _engine(n, gates, len(names), [op])
    up = rng.normal(size=(6, 1)).astype(np.float32)
    vals, grad = eng.expectation_vjp(bits, params, up)
    want, want_grad = C.expectation_vjp(n, gates, params, bits, [op], up)
    np.testing.assert_allclose(vals.cpu().numpy(), want, atol=2e-5 * _op_norm([op])[0])
    np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=1e-4 * max(1.0, np.abs(want_grad).max()))
  ops = [flipflop, xx_only, yy_only, dressed]                                       # weighted: XX and YY of ops 1, 2 share masks
  eng = _engine(n, gates, len(names), ops)
  for up in (rng.normal(size=(6, 4)).astype(np.float32),
             np.tile(np.array([[0.0, 1.0, 1.0, 0.0]], np.float32), (6, 1)),           # ops 1 + 2 cancel like op 0 alone
             np.tile(np.array([[0.0, 1.0, 0.5, 0.0]], np.float32), (6, 1)),
             np.zeros((6, 4), np.float32)):
    vals, grad = eng.expectation_vjp(bits, params, up)
    want, want_grad = C.expectation_vjp(n, gates, params, bits, ops, up)
    np.testing.assert_allclose(vals.cpu().numpy(), want, atol=2e-5 * _op_norm(ops).max())
    np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=1e-4 * max(1.0, np.abs(want_grad).max()))


@pytest.mark.parametrize("n,cut", [(13, 256), (13, 512), (16, 256)])
@pytest.mark.parametrize("kernel", [0, 1])
def test_a_mask_whose_terms_straddle_a_staging_chunk_of_the_gather_kernel(n, cut, kernel):
  """The gather kernel stages 256 terms at a time; a mask whose terms straddle a multiple of 256 in the mask-sorted
  term array is cut into two groups.  The second group must fetch its partners again: the first may have fetched only
  the slot pairs it does not vanish on, or -- XX + YY on two block bits, where the bits agree -- nothing at all (round
  5's advisor finding: the second group then consumed stale partners).  `cut - 2` diagonal strings sort in front of
  XX + YY (+ a Z-dressed XX and YY behind the cut), on masks that leave the block and on one inside it; value mode
  (forward only), lambda mode (VJP) and several observables; the block kernel on the same operator for comparison."""
  from oracle import qhbm_cpu as C
  rng = np.random.default_rng(7 * n + cut)
  gates, names = O.hea_gates(n, 2, "s")
  params = rng.uniform(-1, 1, len(names)).astype(np.float32)
  seen, diag = set(), []
  while len(diag) < cut - 2:
    z = int(rng.integers(1, 1 << n))
    if z not in seen:
      seen.add(z)
      diag.append((float(rng.normal()) * 0.05, 0, z))
  bits = _random_bits(rng, 6, n)
  for a, b in [(0, 1), (0, n // 2), (n - 3, n - 2)]:
    # sorted by mask: the diagonal strings (x = 0), then [XX, YY | XXZ, YYZ, XX'] of the one flip mask
    tail = [O.pauli_term(1.0, [(a, "X"), (b, "X")]), O.pauli_term(1.0, [(a, "Y"), (b, "Y")]),
            O.pauli_term(0.7, [(a, "X"), (b, "X"), (5, "Z")]), O.pauli_term(-0.4, [(a, "Y"), (b, "Y"), (5, "Z")]),
            O.pauli_term(0.3, [(a, "X"), (b, "X"), (n - 1, "Z")])]
    op = diag + tail
    eng = _engine(n, gates, len(names), [op], observable_kernel=kernel)
    norm = _op_norm([op])[0]
    got = eng.expectation(bits, params).cpu().numpy()
    up = rng.normal(size=(6, 1)).astype(np.float32)
    want, want_grad = C.expectation_vjp(n, gates, params, bits, [op], up)
    np.testing.assert_allclose(got, want, atol=2e-5 * norm, err_msg=f"values, pair {(a, b)}")
    vals, grad = eng.expectation_vjp(bits, params, up)
    np.testing.assert_allclose(vals.cpu().numpy(), want, atol=2e-5 * norm)
    np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=1e-4 * max(1.0, np.abs(want_grad).max()),
                               err_msg=f"gradient, pair {(a, b)}")
    # two observables (weighted lambda): the second holds the same masks with other weights
    ops = [op, diag[: cut - 3] + tail[1:] + tail[:1]]
    eng = _engine(n, gates, len(names), ops, observable_kernel=kernel)
    up = rng.normal(size=(6, 2)).astype(np.float32)
    vals, grad = eng.expectation_vjp(bits, params, up)
    want, want_grad = C.expectation_vjp(n, gates, params, bits, ops, up)
    np.testing.assert_allclose(vals.cpu().numpy(), want, atol=2e-5 * _op_norm(ops).max())
    np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=1e-4 * max(1.0, np.abs(want_grad).max()))


def test_results_are_bit_reproducible_and_independent_of_chunking():
  """No floating-point atomics anywhere: expectation values accumulate in 64-bit fixed point,
  gradient partials go wave -> tile -> state in fixed order.  Two runs, and a run cut into chunks
  of 3 states, must agree BIT FOR BIT per state (SURVEY.md section 7 'hard parts': results that do
  not depend on how the batch is sharded)."""
  rng = np.random.default_rng(5)
  n, layers = 15, 3
  gates, names = O.hea_gates(n, layers, "d")
  params = rng.uniform(-1, 1, len(names))
  ops = [O.xxz_chain_op(n), O.tfim_ring_op(n)]
  bits = _random_bits(rng, 8, n)
  eng = _engine(n, gates, len(names), ops, tile_qubits=11, adjoint_tile_qubits=10)
  v1, j1 = eng.expectation_jacobian(bits, params)
  v2, j2 = eng.expectation_jacobian(bits, params)
  assert torch.equal(v1, v2) and torch.equal(j1, j2)
  eng.set_option("chunk_states", 3)
  v3, j3 = eng.expectation_jacobian(bits, params)
  assert torch.equal(v1, v3) and torch.equal(j1, j3)
  # a shard of the batch gives the same rows as the whole batch
  eng.set_option("chunk_states", 0)
  v4, j4 = eng.expectation_jacobian(bits[5:], params)
  assert torch.equal(v1[5:], v4) and torch.equal(j1[5:], j4)
  assert torch.equal(eng.expectation(bits[2:4], params), v1[2:4])


def test_parameter_shift_batches_states_and_programs():
  """The shift VJP runs (state, shifted program) pairs as one batch; whatever the batch geometry --
  all programs at once, three or seven elements per launch set (`chunk_states`), one tile or many --
  it must equal the adjoint VJP and the oracle's shift rule (qnn.py:168; baselines/train.py:190-240)."""
  rng = np.random.default_rng(31)
  n, layers = 11, 2
  gates, names = O.hea_gates(n, layers, "p")
  gates = gates + [(O.GATE_XXPOW, 2, 7, 3, 0.7, 0.1), (O.GATE_ZZPOW, 0, 10, 5, -1.3, 0.0)]   # tied parameters too
  params = rng.uniform(-1, 1, len(names))
  ops = [O.xxz_chain_op(n), O.tfim_ring_op(n)]
  bits = _random_bits(rng, 5, n)
  up = rng.normal(size=(5, 2)).astype(np.float32)
  want_vals, want_jac = O.expectation_jacobian(n, gates, params, bits, ops)
  want = np.einsum("bt,btp->p", up, want_jac)
  tol = 1e-4 * max(1.0, np.abs(want).max())
  for tile in (0, 10):
    for chunk in (0, 3, 7):
      eng = _engine(n, gates, len(names), ops, tile_qubits=tile, chunk_states=chunk)
      vals, grad = eng.expectation_vjp(bits, params, up, method=E.GRAD_PARAMETER_SHIFT)
      np.testing.assert_allclose(vals.cpu().numpy(), want_vals, atol=1e-4)
      np.testing.assert_allclose(grad.cpu().numpy(), want, atol=tol)
      # and the engine is still usable for ordinary calls afterwards
      np.testing.assert_allclose(eng.expectation(bits, params).cpu().numpy(), want_vals, atol=1e-4)


@pytest.mark.parametrize("n,layers,tile,kinds", [(13, 3, 10, "hea"), (14, 2, 10, "all kinds"), (15, 4, 11, "hea"), (16, 2, 0, "hea")])
def test_parameter_shift_programs_share_the_base_program_s_prefix_bit_for_bit(n, layers, tile, kinds):
  """A shifted program differs from the base program in ONE gate: it starts at the first pass that reads that gate's
  coefficients, from the base program's state (`shift_prefix_sharing`, default on; round 5's review, item 2 i).  The
  passes it skips would have computed the same bits, so gradient and values must EQUAL the unshared run bit for bit --
  values taken by the observable kernel (a wide random Pauli sum) and measured in the passes (TFIM, shards), tied
  parameters, a gradient mask, launch sets cut by `chunk_states` -- and agree with the oracle's adjoint VJP."""
  from oracle import qhbm_cpu as C
  rng = np.random.default_rng(700 + n)
  if kinds == "hea":
    gates, names = O.hea_gates(n, layers, "ps")
    n_params = len(names)
    gates = gates + [(O.GATE_XXPOW, 2, n - 3, 3, 0.7, 0.1), (O.GATE_ZZPOW, 0, n - 1, 5, -1.3, 0.0)]   # tied parameters
  else:
    n_params = 9
    gates = random_circuit(rng, n, 60, n_params, kinds=[k for k in range(12) if k != O.GATE_ISWAPPOW])
  params = rng.uniform(-1, 1, n_params).astype(np.float32)
  bits = _random_bits(rng, 3, n)
  layouts = {"wide sum": [O.random_pauli_op(n, 40, n, p_identity=0.6)],
             "tfim + xxz": [O.tfim_ring_op(n), O.xxz_chain_op(n)],
             "shards": [[(1.0, 0, 1 << q)] for q in range(n)] + [[(0.5, 0, (1 << q) | (1 << ((q + 5) % n)))] for q in range(n)]}
  mask = rng.random(n_params) < 0.6
  for name, ops in layouts.items():
    up = rng.normal(size=(3, len(ops))).astype(np.float32)
    want_vals, want = C.expectation_vjp(n, gates, params, bits, ops, up)
    for use_mask in (False, True):
      for chunk in (0, 5):
        results = []
        for sharing in (0, 1):
          eng = _engine(n, gates, n_params, ops, tile_qubits=tile, chunk_states=chunk, shift_prefix_sharing=sharing)
          if use_mask:
            eng.set_gradient_mask(mask)
          vals, grad = eng.expectation_vjp(bits, params, up, method=E.GRAD_PARAMETER_SHIFT)
          results.append((vals.clone(), grad.clone()))
          if sharing:
            fwd, _ = eng.num_passes()
            assert fwd > 1 or tile == 0, "the case is meant to run several forward passes"
        assert torch.equal(results[0][0], results[1][0]), (name, use_mask, chunk)
        assert torch.equal(results[0][1], results[1][1]), (name, use_mask, chunk,
                                                         float((results[0][1] - results[1][1]).abs().max()))
        wg = np.where(mask, want, 0.0) if use_mask else want
        np.testing.assert_allclose(results[1][1].cpu().numpy(), wg, atol=3e-4 * max(1.0, float(np.abs(wg).max())), rtol=0,
                                   err_msg=f"{name} mask={use_mask} chunk={chunk}")
        np.testing.assert_allclose(results[1][0].cpu().numpy(), want_vals, atol=5e-5 * _op_norm(ops).max())


def test_random_circuits_of_every_gate_kind_at_18_qubits_against_c_oracle():
  """The general kernel variants (Y, H, CNOT / SWAP / ISWAP / XX / YY / ZZ powers: dense two-qubit
  ops on the LDS tiles, the two-tile adjoint layout) on multi-pass plans at a size where only the C
  restatement is quick enough to be the checker: values and the VJP of three random circuits."""
  from oracle import qhbm_cpu as C
  n, n_params = 18, 10
  for seed in range(3):
    rng = np.random.default_rng(1800 + seed)
    gates = random_circuit(rng, n, 90, n_params)
    params = rng.uniform(-1, 1, n_params).astype(np.float32)
    ops = [O.random_pauli_op(n, 12, seed, p_identity=0.7), O.xxz_chain_op(n)]
    bits = _random_bits(rng, 3, n)
    up = rng.normal(size=(3, 2)).astype(np.float32)
    want_vals, want_grad = C.expectation_vjp(n, gates, params, bits, ops, up)
    for opts in ({}, {"tile_qubits": 11, "adjoint_tile_qubits": 11}):
      eng = _engine(n, gates, n_params, ops, **opts)
      vals, grad = eng.expectation_vjp(bits, params, up)
      np.testing.assert_allclose(vals.cpu().numpy(), want_vals, atol=5e-5 * max(1.0, _op_norm(ops).max()))
      np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=2e-4 * max(1.0, np.abs(want_grad).max()))


def test_a_compute_call_is_asynchronous_and_graph_capturable():
  """After the first call on a model the engine issues no host copy, allocation or synchronisation:
  a VJP call can be captured into a HIP graph on the caller's stream and replayed with new inputs in
  the same buffers (DESIGN.md 'HIP streams and graphs'); the replay equals an eager call bit for bit."""
  rng = np.random.default_rng(12)
  n, layers = 12, 3
  gates, names = O.hea_gates(n, layers, "g")
  ops = [O.tfim_ring_op(n)]
  eng = _engine(n, gates, len(names), ops)
  bits = torch.from_numpy(_random_bits(rng, 64, n)).cuda()
  params = torch.from_numpy(rng.uniform(-1, 1, len(names)).astype(np.float32)).cuda()
  up = torch.full((64, 1), 1.0 / 64, device="cuda")
  eng.expectation_vjp(bits, params, up)       # first call: plans, uploads, workspace
  graph = torch.cuda.CUDAGraph()
  side = torch.cuda.Stream()
  with torch.cuda.stream(side):
    eng.expectation_vjp(bits, params, up)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
      g_vals, g_grad = eng.expectation_vjp(bits, params, up)
  torch.cuda.synchronize()
  new_params = torch.from_numpy(rng.uniform(-1, 1, len(names)).astype(np.float32)).cuda()
  params.copy_(new_params)                    # same buffer, new values: the graph reads them at replay
  graph.replay()
  torch.cuda.synchronize()
  vals, grad = eng.expectation_vjp(bits, params, up)
  assert torch.equal(g_vals, vals) and torch.equal(g_grad, grad)
  want = O.expectation(n, gates, new_params.cpu().numpy().astype(np.float64), bits.cpu().numpy(), ops)
  np.testing.assert_allclose(vals.cpu().numpy(), want, atol=1e-4)


@pytest.mark.parametrize("mode", ["forward", "vjp", "retained pair"])
def test_a_captured_call_replays_correctly_on_another_stream_than_it_was_warmed_up_on(mode):
  """Round 6: the engine used to zero its accumulators with hipMemsetAsync, which a capture turns into a memset NODE;
  a graph warmed up on a side stream (the `torch.cuda.graph` recipe) and replayed on the default stream then returned
  values offset by a constant from the SECOND replay on (the first was right).  The zero fills are kernels now
  (kernels.hip launch_zero_fill): every replay, on either stream, equals the eager call bit for bit."""
  rng = np.random.default_rng(6)
  n, layers = 6, 2
  gates, names = O.hea_gates(n, layers, "cg")
  eng = _engine(n, gates, len(names), [O.tfim_ring_op(n)])
  bits = torch.from_numpy(_random_bits(rng, 32, n)).cuda()
  params = torch.from_numpy(rng.uniform(-1, 1, len(names)).astype(np.float32)).cuda()
  up = torch.full((32, 1), 1.0 / 32, device="cuda")

  def call():
    if mode == "forward":
      return eng.expectation(bits, params), torch.zeros(1, device="cuda")
    if mode == "vjp":
      return eng.expectation_vjp(bits, params, up)
    vals = eng.expectation(bits, params, retain=True)
    return vals, eng.expectation_vjp_retained(bits, params, up)

  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    for _ in range(2):
      want_vals, want_grad = (t.clone() for t in call())
  torch.cuda.current_stream().wait_stream(side)
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph):
    got_vals, got_grad = call()
  for replay in range(4):
    if replay == 2:
      with torch.cuda.stream(side):
        graph.replay()
    else:
      graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got_vals, want_vals) and torch.equal(got_grad, want_grad), (mode, replay)


def test_x_exponents_far_outside_one_period():
  """X**t is applied as three shears of the exponent reduced to one period (kernels.hip x_pair4,
  prep_coefs_kernel): values, Jacobian and the exported state (global phase included) must not
  notice the reduction, at the period boundaries either."""
  n, P = 12, 5
  rng = np.random.default_rng(77)
  gates = []
  for q in range(n):
    gates.append((E.GATE_XPOW, q, -1, q % P, float(rng.uniform(-3, 3)), float(rng.uniform(-9, 9))))
    gates.append((E.GATE_ZPOW, q, -1, (q + 1) % P, 0.7, 0.1 * q))
  for q in range(n - 1):
    gates.append((E.GATE_CZPOW, q, q + 1, (q + 2) % P, 1.3, -0.2))
  for q, t in zip(range(n), [1.0, -1.0, 2.0, 3.0, -3.0, 0.5, -0.5, 1.5, 4.0, 0.0, 5.0, -7.0]):
    gates.append((E.GATE_XPOW, q, -1, -1, 0.0, t))   # constants on / next to the period boundaries
  for q in range(n):
    gates.append((E.GATE_XPOW, q, -1, (q + 3) % P, -2.5, 6.0))
  params = rng.uniform(-2, 2, P)
  bits = _random_bits(rng, 3, n)
  ops = [O.xxz_chain_op(n), O.random_pauli_op(n, 12, 5)]
  eng = _engine(n, gates, P, ops)
  check_values(eng, n, gates, params, bits, ops, rel=2e-5)
  check_jacobian(eng, n, gates, params, bits[:2], ops, rel=2e-4)
  sv = eng.statevector(bits[:1], params).cpu().numpy()[0]
  np.testing.assert_allclose(sv, O.simulate(n, gates, params, list(bits[0])).ravel(), atol=3e-6)


@pytest.mark.parametrize("tile", [10, 11])
@pytest.mark.parametrize("layout", ["plain", "relabel", "tail-tiles"])
def test_deep_chain_with_and_without_the_scheduler_layout_choices(layout, tile):
  """A circuit deep enough for the adjoint tail -- bits with no gate left -- against the oracle under the
  three layouts the scheduler has: "relabel" (default: the pass that finishes an index bit stores its
  tiles with that bit moved out of the 128-byte lines, later passes load whole live lines and clear the
  stale half where they hold a moved bit), "tail-tiles" (`adjoint_relabel` = 0: dead waves (tile 11: one
  wave bit), tiles without the low index bits (tile 10), pruning on every finished non-local bit) and
  "plain" (`cph_wave_bits` = 0, kept for A/B measurements); the engines must also agree with each other
  far inside the oracle tolerance."""
  n, layers = 15, 10
  rng = np.random.default_rng(4242)
  gates, names = O.hea_gates(n, layers, "deep")
  params = rng.uniform(-1, 1, len(names))
  ops = [O.xxz_chain_op(n)]
  bits = _random_bits(rng, 2, n)
  opts = dict(tile_qubits=tile, adjoint_tile_qubits=tile)
  layout_opts = {"plain": dict(cph_wave_bits=0), "relabel": {}, "tail-tiles": dict(adjoint_relabel=0)}
  eng = _engine(n, gates, len(names), ops, **layout_opts[layout], **opts)
  fwd, bwd = eng.num_passes()
  assert fwd >= 4 and bwd >= 3
  adjoint = eng.describe_schedule()
  adjoint = adjoint[adjoint.index("adjoint"):]
  dead = [tok for line in adjoint.splitlines() if "dead=" in line for tok in line.split("dead=")[1].split(",")]
  if layout == "relabel":
    assert adjoint.startswith("adjoint (relabeling) plan") and "moves-local-bits=" in adjoint
    assert " c=0 " not in adjoint              # every tile is made of whole lines
  elif layout == "tail-tiles":
    assert adjoint.startswith("adjoint plan") and "moves-local-bits=" not in adjoint
    if tile == 10:
      assert " c=0 " in adjoint                # a tail pass without the low index bits
    else:
      assert any(tok != "0" for tok in dead)   # rounds in which some waves hold zeros only
  else:
    assert adjoint.startswith("adjoint plan") and "moves-local-bits=" not in adjoint
  check_values(eng, n, gates, params, bits, ops, rel=3e-5)
  want_jac = check_jacobian(eng, n, gates, params, bits[:1], ops, rel=3e-4)
  other = _engine(n, gates, len(names), ops, **layout_opts["plain" if layout != "plain" else "tail-tiles"], **opts)
  _, jac = other.expectation_jacobian(bits[:1], params)
  _, mine = eng.expectation_jacobian(bits[:1], params)
  np.testing.assert_allclose(mine.cpu().numpy(), jac.cpu().numpy(), atol=2e-5 * max(1.0, np.abs(want_jac).max()), rtol=0)


@pytest.mark.parametrize("n,tile", [(11, 0), (14, 11)])
def test_single_observable_values_from_lambda_match_the_measured_ones(n, tile):
  """With one observable the fused value + VJP call and the retained forward take <psi|O|psi> from
  lambda = O psi and measure nothing in the forward sweep (`values_from_observable`, default on).  Values
  and gradients must equal those of the measuring path and the oracle -- with zero and negative upstream
  weights too (lambda is computed unweighted, the weight goes onto the gradient row)."""
  rng = np.random.default_rng(900 + n)
  gates, names = O.hea_gates(n, 3, "vm")
  params = rng.uniform(-1, 1, len(names))
  op = O.random_pauli_op(n, 20, 3) + O.xxz_chain_op(n)
  bits = _random_bits(rng, 4, n)
  up = np.array([[0.7], [0.0], [-1.3], [2.0]], np.float32)
  opts = dict(tile_qubits=tile, adjoint_tile_qubits=max(tile - 1, 0)) if tile else {}
  want_vals, want_jac = O.expectation_jacobian(n, gates, params, bits, [op])
  want_grad = np.einsum("bt,btp->p", up, want_jac)
  results = []
  for flag in (1, 0):
    eng = _engine(n, gates, len(names), [op], values_from_observable=flag, **opts)
    vals, grad = eng.expectation_vjp(bits, params, up)
    np.testing.assert_allclose(vals.cpu().numpy(), want_vals, atol=2e-5 * _op_norm([op]).max(), rtol=0)
    np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=2e-4 * max(1.0, np.abs(want_grad).max()), rtol=0)
    # retained forward, then the backward from the kept (psi, O psi) pair
    rv = eng.expectation(bits, params, retain=True)
    assert eng.retained is not None
    np.testing.assert_allclose(rv.cpu().numpy(), want_vals, atol=2e-5 * _op_norm([op]).max(), rtol=0)
    rg = eng.expectation_vjp_retained(bits, params, up)
    np.testing.assert_allclose(rg.cpu().numpy(), want_grad, atol=2e-4 * max(1.0, np.abs(want_grad).max()), rtol=0)
    results.append((vals.cpu().numpy(), grad.cpu().numpy()))
  np.testing.assert_allclose(results[0][0], results[1][0], atol=5e-6 * _op_norm([op]).max(), rtol=0)
  np.testing.assert_allclose(results[0][1], results[1][1], atol=5e-5 * max(1.0, np.abs(want_grad).max()), rtol=0)


@pytest.mark.parametrize("n,layers,adj_tile", [(14, 6, 0), (15, 14, 10), (16, 14, 10), (17, 6, 11), (17, 20, 11), (15, 20, 10)])
def test_adjoint_plans_rebuilt_from_a_candidate_order(n, layers, adj_tile):
  """`adjoint_plan_search` (default on) rebuilds the backward plan from the scheduler's candidate pass orders and
  keeps the one with the least modelled time: on these circuits that is NOT the scheduler's first choice.  Values
  and the VJP of both plans against the C oracle, and against each other."""
  from oracle import qhbm_cpu as C
  rng = np.random.default_rng(31 * n + layers)
  gates, names = O.hea_gates(n, layers, "ps")
  params = rng.uniform(-1, 1, len(names)).astype(np.float32)
  ops = [O.tfim_ring_op(n)]
  bits = _random_bits(rng, 5, n)
  up = rng.normal(size=(5, 1)).astype(np.float32)
  want, want_grad = C.expectation_vjp(n, gates, params, bits, ops, up)
  plans, grads = [], []
  for search in (0, 1):
    eng = _engine(n, gates, len(names), ops, adjoint_tile_qubits=adj_tile, adjoint_plan_search=search)
    text = eng.describe_schedule()
    plans.append(text[text.index("adjoint"):])
    vals, grad = eng.expectation_vjp(bits, params, up)
    np.testing.assert_allclose(vals.cpu().numpy(), want, atol=2e-5 * _op_norm(ops)[0])
    np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=1e-4 * max(1.0, np.abs(want_grad).max()))
    grads.append(grad.cpu().numpy())
  assert plans[0] != plans[1]
  np.testing.assert_allclose(grads[0], grads[1], atol=2e-5 * max(1.0, np.abs(want_grad).max()))


@pytest.mark.parametrize("n,layers,adj_tile", [(6, 3, 0), (13, 4, 10), (15, 6, 11), (16, 8, 0)])
def test_gradient_mask_freezes_parameters_and_shortens_the_backward_sweep(n, layers, adj_tile):
  """`qhbm_set_gradient_mask`: frozen parameters get zero gradient entries, the others are unchanged -- for the
  adjoint sweep (which stops at the first gate, in circuit order, of a live parameter and then prunes nothing: what
  is left of psi is not a basis state) and for the shift rule (no programs for frozen gates).  Masks: the leading
  half of the circuit (the QMHL case: a fixed data circuit in front of U_model^dagger), leading diagonal gates only,
  scattered parameters, everything, nothing; against the C oracle's full gradient."""
  from oracle import qhbm_cpu as C
  rng = np.random.default_rng(100 * n + layers)
  hea, names = O.hea_gates(n, layers, "gm")
  P = len(names) + 4                      # four leading diagonal gates with parameters of their own, then the HEA
  gates = [(E.GATE_ZPOW, q, -1, len(names) + q, 1.0, 0.1) for q in range(3)] + [(E.GATE_CZPOW, 0, 1, len(names) + 3, -0.5, 0.0)] + hea
  params = rng.uniform(-1, 1, P).astype(np.float32)
  ops = [O.xxz_chain_op(n), O.tfim_ring_op(n)]
  bits = _random_bits(rng, 4, n)
  up = rng.normal(size=(4, 2)).astype(np.float32)
  want, want_grad = C.expectation_vjp(n, gates, params, bits, ops, up)
  tol = 1e-4 * max(1.0, np.abs(want_grad).max())
  first_half = {g[3] for g in gates[:len(gates) // 2] if g[3] >= 0}
  first_x_layer = {g[3] for g in gates[:n + 4] if g[3] >= 0}
  lead_diag = set()
  for g in gates:                         # the parameters of the gates before the first parametrised X**t, if any are diagonal
    if g[3] >= 0 and g[0] == E.GATE_XPOW:
      break
    if g[3] >= 0:
      lead_diag.add(g[3])
  masks = {"leading half": np.array([p not in first_half for p in range(P)]),
           "first X layer": np.array([p not in first_x_layer for p in range(P)]),
           "scattered": rng.random(P) < 0.5,
           "everything frozen": np.zeros(P, bool),
           "nothing frozen": np.ones(P, bool)}
  assert len(lead_diag) == 4
  masks["leading diagonal gates"] = np.array([p not in lead_diag for p in range(P)])
  for stop_early, (name, mask) in itertools.product((1, 0, -1), masks.items()):
    eng = _engine(n, gates, P, ops, adjoint_tile_qubits=adj_tile, adjoint_stop_early=stop_early)
    eng.set_gradient_mask(mask)
    vals, grad = eng.expectation_vjp(bits, params, up)
    np.testing.assert_allclose(vals.cpu().numpy(), want, atol=2e-5 * _op_norm(ops).max(), err_msg=name)
    got = grad.cpu().numpy()
    assert (got[~mask] == 0).all(), name
    np.testing.assert_allclose(got[mask], want_grad[mask], atol=tol, err_msg=name)
    if name == "leading half" and stop_early < 0:  # (stopping early forgoes the pruning of the tail: the cheaper plan is kept)
      assert eng.flop_model(1, True)["bwd_flops"] <= eng_full_flops(n, gates, P, ops, adj_tile)
    if name == "leading half" and stop_early == 1 and n > 12:
      assert "adjoint" in eng.describe_schedule() and eng.num_passes()[1] >= 1
    if name == "everything frozen":
      assert eng.num_passes()[1] == 0
    # retained forward + backward from the kept pair, and per-state rows
    rv = eng.expectation(bits, params, retain=True)
    rg = eng.expectation_vjp_retained(bits, params, up).cpu().numpy()
    np.testing.assert_allclose(rg[mask], want_grad[mask], atol=tol, err_msg=name)
    assert (rg[~mask] == 0).all()
    if n <= 13:                                              # the shift rule: 2 forwards per live gate occurrence
      _, sg = eng.expectation_vjp(bits, params, up, E.GRAD_PARAMETER_SHIFT)
      sg = sg.cpu().numpy()
      assert (sg[~mask] == 0).all(), name
      np.testing.assert_allclose(sg[mask], want_grad[mask], atol=3 * tol, err_msg=name)
  eng.set_gradient_mask(None)
  _, grad = eng.expectation_vjp(bits, params, up)
  np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=tol)
  with pytest.raises(E.EngineError, match="gradient mask"):
    eng.set_gradient_mask(np.ones(P + 1, bool))
  eng.set_circuit(n, gates, P)                               # a new circuit forgets the mask
  eng.set_observables(ops)
  _, grad = eng.expectation_vjp(bits, params, up)
  np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=tol)


def eng_full_flops(n, gates, n_params, ops, adj_tile):
  return _engine(n, gates, n_params, ops, adjoint_tile_qubits=adj_tile).flop_model(1, True)["bwd_flops"]


@pytest.mark.parametrize("n,tile,layers", [(10, 0, 2), (12, 0, 2), (13, 10, 3), (15, 11, 3), (16, 12, 2)])
def test_many_diagonal_terms_through_the_walsh_hadamard_measurement(n, tile, layers, monkeypatch):
  """Diagonal groups of >= 32 terms -- the Z-string shards of a modular Hamiltonian (energy.py:165-167, 200-209) --
  are measured through ONE Walsh-Hadamard transform of the tile's |psi|^2 (program.h OP_MEASURE_WHT): every term is
  a single coefficient of it.  KOBE-2 shards (one term per operator), Z strings of every weight with bits among the
  register, lane, wave and tile bits, one operator of many diagonal terms next to non-diagonal operators, and more
  operators than the scratch leaves room for (fallback to the per-term path); against the C oracle and against the
  per-term path (QHBM_NO_WHT), forward-only, chunked, with gradients, and under the batched shift rule."""
  from oracle import qhbm_cpu as C
  rng = np.random.default_rng(17 * n + tile)
  gates, names = O.hea_gates(n, layers, "wh")
  params = rng.uniform(-1, 1, len(names)).astype(np.float32)
  bits = _random_bits(rng, 4, n)
  opts = dict(tile_qubits=tile) if tile else {}
  shards = O.kobe_shards(n, 2)
  strings = [[O.pauli_term(float(rng.normal()), [(int(q), "Z") for q in rng.choice(n, size=int(rng.integers(1, n + 1)), replace=False)])]
             for _ in range(70)]
  dense_sum = [t for op in strings[:50] for t in op]
  cases = {"kobe-2 shards": shards,
           "z strings of every weight": strings,
           "one operator of 50 diagonal terms beside XXZ and TFIM": [dense_sum, O.xxz_chain_op(n), O.tfim_ring_op(n)] + shards[:n]}
  if n <= 13:
    cases["more operators than the scratch leaves"] = (shards * (900 // len(shards) + 1))[:900]
  for name, ops in cases.items():
    up = rng.normal(size=(4, len(ops))).astype(np.float32)
    want, want_grad = C.expectation_vjp(n, gates, params, bits, ops, up)
    tol = 3e-5 * np.maximum(_op_norm(ops), 1.0)
    got = {}
    for no_wht in ("", "1"):
      if no_wht:
        monkeypatch.setenv("QHBM_NO_WHT", "1")
      else:
        monkeypatch.delenv("QHBM_NO_WHT", raising=False)
      eng = _engine(n, gates, len(names), ops, **opts)
      got[no_wht] = eng.expectation(bits, params).cpu().numpy()
      assert (np.abs(got[no_wht] - want) <= tol[None, :]).all(), (name, no_wht, np.abs(got[no_wht] - want).max())
      eng.set_option("chunk_states", 2)
      np.testing.assert_array_equal(eng.expectation(bits, params).cpu().numpy(), got[no_wht])
      if not no_wht and len(ops) <= 200:
        # value + gradient calls measure the same way (several operators: the values come from the passes)
        eng.set_option("chunk_states", 0)
        vals, grad = eng.expectation_vjp(bits, params, up)
        np.testing.assert_array_equal(vals.cpu().numpy(), got[no_wht])
        np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=1e-4 * max(1.0, np.abs(want_grad).max()))
        if n <= 12 and name == "kobe-2 shards":                # batched shifted programs share the measurement code
          _, sg = eng.expectation_vjp(bits, params, up, E.GRAD_PARAMETER_SHIFT)
          np.testing.assert_allclose(sg.cpu().numpy(), want_grad, atol=3e-4 * max(1.0, np.abs(want_grad).max()))
    monkeypatch.delenv("QHBM_NO_WHT", raising=False)
    np.testing.assert_allclose(got[""], got["1"], atol=3e-6 * max(1.0, _op_norm(ops).max()), err_msg=name)


@pytest.mark.parametrize("n,tile", [(5, 0), (11, 0), (13, 10), (14, 11)])
def test_cnot_and_hadamard_sandwiches_are_fused_into_lean_rotations(n, tile, monkeypatch):
  """tfq.util.exponential writes exp(-i theta Z_a Z_b / 2) as CNOT rz CNOT and exp(-i theta X / 2) as H rz H
  (circuit.py:268-272).  The scheduler lowers such adjacent triples to ZZ**t / X**t of the ROTATION's gate (exact,
  global phase included), and leaves look-alikes alone: other qubits, an even or fractional or parametrised outer
  exponent, a rotation on the control.  Values, adjoint and shift-rule gradients and the exported state (phase!)
  against the oracle, with the fusion on and off (QHBM_NO_SANDWICH_FUSION)."""
  rng = np.random.default_rng(50 + n)
  P = 6
  gates = []
  def rz(q, p=None):
    p = int(rng.integers(P)) if p is None else p
    return (E.GATE_ZPOW, q, -1, p, float(rng.uniform(0.3, 1.2)), float(rng.uniform(-0.3, 0.3)), -0.5)
  for layer in range(3):
    for q in range(n):                                         # mixing layer: H rz H, forward and inverted exponents
      sign = 1.0 if (q + layer) % 2 else -1.0
      gates += [(E.GATE_HPOW, q, -1, -1, 0.0, sign), rz(q), (E.GATE_HPOW, q, -1, -1, 0.0, -sign)]
    for a in range(n):                                         # ZZ ring: CNOT rz CNOT
      b = (a + 1) % n
      gates += [(E.GATE_CNOTPOW, a, b, -1, 0.0, 1.0), rz(b), (E.GATE_CNOTPOW, a, b, -1, 0.0, -1.0 if a % 2 else 3.0)]
    gates.append((E.GATE_XPOW, layer % n, -1, layer, 1.0, 0.1))
  look_alikes = [
      [(E.GATE_CNOTPOW, 0, 1, -1, 0.0, 1.0), rz(0, 1), (E.GATE_CNOTPOW, 0, 1, -1, 0.0, 1.0)],          # rotation on the control
      [(E.GATE_CNOTPOW, 0, 1, -1, 0.0, 1.0), rz(1, 2), (E.GATE_CNOTPOW, 1, 0, -1, 0.0, 1.0)],          # reversed second CNOT
      [(E.GATE_CNOTPOW, 0, 1, -1, 0.0, 2.0), rz(1, 3), (E.GATE_CNOTPOW, 0, 1, -1, 0.0, 1.0)],          # even exponent
      [(E.GATE_CNOTPOW, 0, 1, -1, 0.0, 0.5), rz(1, 4), (E.GATE_CNOTPOW, 0, 1, -1, 0.0, -0.5)],         # fractional
      [(E.GATE_CNOTPOW, 0, 1, 5, 1.0, 0.0), rz(1, 0), (E.GATE_CNOTPOW, 0, 1, 5, -1.0, 0.0)],           # parametrised
      [(E.GATE_HPOW, 2, -1, -1, 0.0, 1.0), rz(3, 1), (E.GATE_HPOW, 2, -1, -1, 0.0, 1.0)],             # rotation elsewhere
      [(E.GATE_HPOW, 2, -1, -1, 0.0, 0.5), rz(2, 2), (E.GATE_HPOW, 2, -1, -1, 0.0, -0.5)],            # fractional H
      [(E.GATE_HPOW, 4, -1, -1, 0.0, 1.0), rz(4, 3), (E.GATE_HPOW, 4, -1, -1, 0.0, 1.0), rz(4, 4), (E.GATE_HPOW, 4, -1, -1, 0.0, 1.0)],  # overlapping triples
  ]
  for extra in look_alikes:
    gates += extra
  params = rng.uniform(-1, 1, P)
  ops = [O.xxz_chain_op(n), O.tfim_ring_op(n)]
  bits = _random_bits(rng, 3, n)
  opts = dict(tile_qubits=tile, adjoint_tile_qubits=tile) if tile else {}
  want_vals, want_jac = O.expectation_jacobian(n, gates, params, bits, ops)
  up = rng.normal(size=(3, 2)).astype(np.float32)
  want_grad = np.einsum("bt,btp->p", up, want_jac)
  passes = {}
  for off in ("", "1"):
    if off:
      monkeypatch.setenv("QHBM_NO_SANDWICH_FUSION", "1")
    else:
      monkeypatch.delenv("QHBM_NO_SANDWICH_FUSION", raising=False)
    eng = _engine(n, gates, P, ops, **opts)
    text = eng.describe_schedule()
    passes[off] = sum(int(x) for x in __import__("re").findall(r"mat_ops=(\d+)", text[:text.index("adjoint")]))
    vals, grad = eng.expectation_vjp(bits, params, up)
    np.testing.assert_allclose(vals.cpu().numpy(), want_vals, atol=2e-5 * _op_norm(ops).max())
    np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=2e-4 * max(1.0, np.abs(want_grad).max()))
    _, sg = eng.expectation_vjp(bits, params, up, E.GRAD_PARAMETER_SHIFT)
    np.testing.assert_allclose(sg.cpu().numpy(), want_grad, atol=5e-4 * max(1.0, np.abs(want_grad).max()))
    states = eng.statevector(bits, params).cpu().numpy()
    for row, b in zip(states, bits):
      np.testing.assert_allclose(row, O.simulate(n, gates, params, list(b)).ravel(), atol=5e-6)
  monkeypatch.delenv("QHBM_NO_SANDWICH_FUSION", raising=False)
  assert passes[""] < passes["1"] / 2        # (one-qubit gates of the forward plan: a literal CNOT / H costs X**(1/2)s of its own)


@pytest.mark.parametrize("n,tile", [(12, 10), (15, 11)])
def test_forward_only_values_from_the_observable_kernel(n, tile):
  """A forward-only call with one observable on a multi-pass plan takes <psi|O|psi> from the lambda = O psi
  kernel (nothing stored) when some term flips two or more qubits (`forward_values_from_observable`: auto),
  and keeps measuring in the tiles for single-flip + diagonal sums (TFIM) and on single-pass plans.  Both
  routes against the oracle and against each other, chunked too; several observables: see the end."""
  rng = np.random.default_rng(70 + n)
  gates, names = O.hea_gates(n, 3, "fo")
  params = rng.uniform(-1, 1, len(names))
  bits = _random_bits(rng, 5, n)
  xxz, tfim = O.xxz_chain_op(n), O.tfim_ring_op(n)
  launches = {}
  for name, op in (("xxz", xxz), ("tfim", tfim)):
    want = O.expectation(n, gates, params, bits, [op])
    got = {}
    for flag in (-1, 0, 1):
      eng = _engine(n, gates, len(names), [op], tile_qubits=tile, forward_values_from_observable=flag, profile_events=1)
      assert eng.num_passes()[0] > 1
      got[flag] = eng.expectation(bits, params).cpu().numpy()
      launches[name, flag] = eng.kernel_time_ms()["obs_launches"]
      np.testing.assert_allclose(got[flag], want, atol=1e-5 * _op_norm([op])[0], rtol=0)
      eng.set_option("chunk_states", 2)
      np.testing.assert_array_equal(eng.expectation(bits, params).cpu().numpy(), got[flag])
    np.testing.assert_allclose(got[0], got[1], atol=3e-6 * _op_norm([op])[0], rtol=0)
  assert launches["xxz", -1] == 1 and launches["xxz", 0] == 0 and launches["xxz", 1] == 1
  assert launches["tfim", -1] == 0 and launches["tfim", 1] == 1
  single = _engine(min(n, 12), *O.hea_gates(min(n, 12), 2, "fo")[:1], len(O.hea_gates(min(n, 12), 2, "fo")[1]), [O.xxz_chain_op(min(n, 12))],
                   profile_events=1)
  if single.num_passes()[0] == 1:                      # one tile: the state never reaches HBM
    single.expectation(_random_bits(rng, 3, min(n, 12)), rng.uniform(-1, 1, len(O.hea_gates(min(n, 12), 2, "fo")[1])))
    assert single.kernel_time_ms()["obs_launches"] == 0
  # several observables: measured in the passes below 13 qubits and whenever `multi_observable_values` is off; from
  # ONE launch of the block kernel (csrc/observable.hip) otherwise -- some term flips two qubits
  both = _engine(n, gates, len(names), [xxz, tfim], tile_qubits=tile, forward_values_from_observable=1, profile_events=1)
  check_values(both, n, gates, params, bits, [xxz, tfim])
  assert both.kernel_time_ms()["obs_launches"] == (1 if n >= 13 else 0)
  measured = _engine(n, gates, len(names), [xxz, tfim], tile_qubits=tile, multi_observable_values=0, profile_events=1)
  check_values(measured, n, gates, params, bits, [xxz, tfim])
  assert measured.kernel_time_ms()["obs_launches"] == 0


# ---- round-3 layouts: relabeling adjoint plans, compact grids, paired forward passes ---------------
@pytest.mark.parametrize("n,layers,tile,states,seed", [(13, 6, 10, 5, 1), (14, 9, 10, 4, 2), (15, 8, 11, 3, 3),
                                                        (16, 12, 12, 7, 4), (17, 14, 12, 2, 5), (16, 5, 11, 1, 6)])
def test_lean_chain_circuits_under_every_layout_switch(n, layers, tile, states, seed):
  """Lean circuits (X / Z / CZ / ZZ powers, constant and parametrised, on a scrambled chain) deep enough
  for finished bits, with ODD and single-state batches: the default engine -- adjoint plans that relabel
  finished index bits, grids of live tiles only, dense forward passes on pairs of states, a first forward
  pass that writes one tile per state -- against the oracle, and against the same engine with each of those
  switched off (`adjoint_relabel`, `forward_pairs`, `cph_wave_bits`) and with the wider last forward pass forced
  on against the explicit tile size (`wide_last_pass`), value + VJP (single observable:
  values from lambda = O psi) and two observables (measured values), chunked and retained."""
  rng = np.random.default_rng(9000 + seed)
  n_params = 7
  perm = rng.permutation(n)
  gates = []
  for layer in range(layers):
    for q in range(n):
      if rng.random() < 0.85:
        gates.append((E.GATE_XPOW, int(perm[q]), -1, int(rng.integers(n_params)) if rng.random() < 0.8 else -1,
                      float(rng.uniform(-1.2, 1.2)), float(rng.uniform(-0.4, 0.4))))
      if rng.random() < 0.7:
        gates.append((E.GATE_ZPOW, int(perm[q]), -1, int(rng.integers(n_params)), float(rng.uniform(-1, 1)), 0.1))
    for q0 in range(layer % 2, n - 1, 2):
      kind = E.GATE_CZPOW if rng.random() < 0.75 else E.GATE_ZZPOW
      gates.append((kind, int(perm[q0]), int(perm[q0 + 1]), int(rng.integers(n_params)) if rng.random() < 0.9 else -1,
                    float(rng.uniform(-1, 1)), float(rng.uniform(-0.3, 0.3))))
  params = rng.uniform(-1, 1, n_params)
  bits = _random_bits(rng, states, n)
  xxz = O.xxz_chain_op(n)
  opts = dict(tile_qubits=tile, adjoint_tile_qubits=min(tile, 12))
  up1 = rng.normal(size=(states, 1)).astype(np.float32)
  want_v, want_jac = O.expectation_jacobian(n, gates, params, bits[:2], [xxz])
  results = {}
  for name, extra in (("default", {}), ("no-relabel", dict(adjoint_relabel=0)), ("no-pairs", dict(forward_pairs=0)),
                      ("plain", dict(cph_wave_bits=0)), ("chunked", dict(chunk_states=2)),
                      ("wide-last-pass", dict(wide_last_pass=1))):
    eng = _engine(n, gates, n_params, [xxz], **opts, **extra)
    vals, grad = eng.expectation_vjp(bits, params, up1)
    rows = eng.state_gradients(states).cpu().numpy()
    results[name] = (vals.cpu().numpy(), grad.cpu().numpy(), rows)
    if name == "default":
      assert "relabeling" in eng.describe_schedule() or eng.num_passes()[1] == 1
      np.testing.assert_allclose(vals.cpu().numpy()[:2], want_v, atol=2e-5 * _op_norm([xxz])[0])
      want_rows = up1[:2] * want_jac[:, 0, :]
      np.testing.assert_allclose(rows[:2], want_rows, atol=2e-4 * max(1.0, np.abs(want_rows).max()))
      # forward now, backward later from the retained states
      eng.expectation(bits, params, retain=True)
      if eng.retained is not None:
        np.testing.assert_allclose(eng.expectation_vjp_retained(bits, params, up1).cpu().numpy(), grad.cpu().numpy(),
                                   atol=1e-6 * max(1.0, float(np.abs(grad.cpu().numpy()).max())))
  scale = max(1.0, np.abs(results["default"][1]).max())
  for name, (v, g, r) in results.items():
    np.testing.assert_allclose(v, results["default"][0], atol=3e-5, err_msg=name)
    np.testing.assert_allclose(g, results["default"][1], atol=3e-5 * scale, err_msg=name)
    np.testing.assert_allclose(r, results["default"][2], atol=3e-5 * scale, err_msg=name)
  # two observables: the forward sweep measures (no paired passes where a pass measures), full Jacobian of two states
  ops2 = [xxz, O.tfim_ring_op(n)]
  eng = _engine(n, gates, n_params, ops2, **opts)
  check_values(eng, n, gates, params, bits, ops2, rel=3e-5)
  check_jacobian(eng, n, gates, params, bits[:1], ops2, rel=3e-4)


@pytest.mark.parametrize("n,tile", [(4, 0), (12, 10), (14, 11)])
def test_constant_hadamards_and_cnots_run_as_lean_ops(n, tile, monkeypatch):
  """A constant H**k / CNOT**k (k odd) is lowered to e^{-i pi/4} Z^(1/2) X^(1/2) Z^(1/2) / H_t CZ H_t (schedule.cpp
  lower()), so Clifford + rotation circuits stay on the lean kernels; parametrised, even or fractional exponents keep
  the dense path.  Values, gradients (adjoint and shift rule) and the exported state with its global phase against
  the oracle, with the lowering on and off (QHBM_NO_LEAN_CLIFFORD); the Bell pair of qhbm_utils_test.py:28-51."""
  rng = np.random.default_rng(9 * n)
  P = 5
  gates = []
  for layer in range(3):
    for q in range(n):
      gates.append((E.GATE_HPOW, q, -1, -1, 0.0, float(rng.choice([1.0, -1.0, 3.0]))))
      gates.append((E.GATE_ZPOW, q, -1, int(rng.integers(P)), float(rng.uniform(0.2, 1.0)), 0.1))
    for a in range(n - 1):
      c, t = (a, a + 1) if (a + layer) % 2 else (a + 1, a)
      gates.append((E.GATE_CNOTPOW, c, t, -1, 0.0, float(rng.choice([1.0, -1.0]))))
      gates.append((E.GATE_XPOW, t, -1, int(rng.integers(P)), 0.7, -0.2))
  gates += [(E.GATE_HPOW, 0, -1, -1, 0.0, 2.0), (E.GATE_HPOW, 1, -1, -1, 0.0, 0.5), (E.GATE_HPOW, 2, -1, 0, 1.0, 0.0),
            (E.GATE_CNOTPOW, 0, 1, -1, 0.0, 0.5), (E.GATE_CNOTPOW, 2, 3, 1, 1.0, 0.0), (E.GATE_CNOTPOW, 3, 2, -1, 0.0, 2.0)]
  params = rng.uniform(-1, 1, P)
  ops = [O.xxz_chain_op(n), O.tfim_ring_op(n)]
  bits = _random_bits(rng, 3, n)
  opts = dict(tile_qubits=tile, adjoint_tile_qubits=tile) if tile else {}
  want_vals, want_jac = O.expectation_jacobian(n, gates, params, bits, ops)
  up = rng.normal(size=(3, 2)).astype(np.float32)
  want_grad = np.einsum("bt,btp->p", up, want_jac)
  for off in ("", "1"):
    if off:
      monkeypatch.setenv("QHBM_NO_LEAN_CLIFFORD", "1")
    else:
      monkeypatch.delenv("QHBM_NO_LEAN_CLIFFORD", raising=False)
    eng = _engine(n, gates, P, ops, **opts)
    vals, grad = eng.expectation_vjp(bits, params, up)
    np.testing.assert_allclose(vals.cpu().numpy(), want_vals, atol=2e-5 * _op_norm(ops).max())
    np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=2e-4 * max(1.0, np.abs(want_grad).max()))
    _, sg = eng.expectation_vjp(bits, params, up, E.GRAD_PARAMETER_SHIFT)
    np.testing.assert_allclose(sg.cpu().numpy(), want_grad, atol=5e-4 * max(1.0, np.abs(want_grad).max()))
    states = eng.statevector(bits, params).cpu().numpy()
    for row, b in zip(states, bits):
      np.testing.assert_allclose(row, O.simulate(n, gates, params, list(b)).ravel(), atol=5e-6)
    bell = E.Engine(0)
    bell.set_circuit(2, [(E.GATE_HPOW, 0, -1, -1, 0.0, 1.0), (E.GATE_CNOTPOW, 0, 1, -1, 0.0, 1.0)], 0)
    got = bell.statevector(np.zeros((1, 2), np.int8), np.zeros(0, np.float32)).cpu().numpy()[0]
    np.testing.assert_allclose(got, np.array([1, 0, 0, 1]) / np.sqrt(2), atol=1e-6)
  monkeypatch.delenv("QHBM_NO_LEAN_CLIFFORD", raising=False)


@pytest.mark.parametrize("n,tile", [(3, 0), (12, 10), (14, 11)])
def test_y_powers_run_as_conjugated_x_powers(n, tile, monkeypatch):
  """Y**t = S X**t S^dagger exactly (cirq's phase included): the scheduler lowers a Y power -- parametrised or not --
  to the X power of the same gate between two fixed phases, and folds those into neighbouring Z**t gates on the same
  qubit (Z^(1/2) Z**t = Z**(t + 1/2)).  ry-style layers, Y next to Y, Y next to H / CNOT, Y at both ends of the circuit;
  values, adjoint and shift-rule gradients and the exported state against the oracle, lowering on and off."""
  rng = np.random.default_rng(3 * n + 1)
  P = 6
  gates = [(E.GATE_YPOW, 0, -1, 0, 1.0, 0.0), (E.GATE_YPOW, 0, -1, 1, -0.5, 0.3)]           # Y Y on one qubit, at the start
  for layer in range(3):
    for q in range(n):
      gates.append((E.GATE_YPOW, q, -1, int(rng.integers(P)), float(rng.uniform(0.3, 1.0)), float(rng.uniform(-0.2, 0.2))))
      gates.append((E.GATE_ZPOW, q, -1, int(rng.integers(P)), float(rng.uniform(0.3, 1.0)), 0.0))
    for a in range(0, n - 1):
      gates.append((E.GATE_CZPOW, a, a + 1, int(rng.integers(P)), 1.0, 0.0))
    gates.append((E.GATE_HPOW, layer % n, -1, -1, 0.0, 1.0))
    gates.append((E.GATE_YPOW, layer % n, -1, -1, 0.0, 0.37))                               # constant Y after an H
    if n > 1:
      gates.append((E.GATE_CNOTPOW, 0, 1, -1, 0.0, 1.0))
  gates.append((E.GATE_YPOW, n - 1, -1, 2, 2.5, -1.7))                                      # far outside one period, at the end
  params = rng.uniform(-1, 1, P)
  ops = [O.xxz_chain_op(n), O.tfim_ring_op(n)] if n > 3 else [O.tfim_ring_op(n), [O.pauli_term(1.0, [(0, "Y")])]]
  bits = _random_bits(rng, 3, n)
  opts = dict(tile_qubits=tile, adjoint_tile_qubits=tile) if tile else {}
  want_vals, want_jac = O.expectation_jacobian(n, gates, params, bits, ops)
  up = rng.normal(size=(3, 2)).astype(np.float32)
  want_grad = np.einsum("bt,btp->p", up, want_jac)
  for off in ("", "1"):
    if off:
      monkeypatch.setenv("QHBM_NO_LEAN_CLIFFORD", "1")
    else:
      monkeypatch.delenv("QHBM_NO_LEAN_CLIFFORD", raising=False)
    eng = _engine(n, gates, P, ops, **opts)
    vals, grad = eng.expectation_vjp(bits, params, up)
    np.testing.assert_allclose(vals.cpu().numpy(), want_vals, atol=2e-5 * _op_norm(ops).max())
    np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=2e-4 * max(1.0, np.abs(want_grad).max()))
    _, sg = eng.expectation_vjp(bits, params, up, E.GRAD_PARAMETER_SHIFT)
    np.testing.assert_allclose(sg.cpu().numpy(), want_grad, atol=5e-4 * max(1.0, np.abs(want_grad).max()))
    states = eng.statevector(bits, params).cpu().numpy()
    for row, b in zip(states, bits):
      np.testing.assert_allclose(row, O.simulate(n, gates, params, list(b)).ravel(), atol=5e-6)
  monkeypatch.delenv("QHBM_NO_LEAN_CLIFFORD", raising=False)


# ---- every gate kind against cirq's DOCUMENTED matrices, end to end (finish the oracle pin) ----------
def test_every_gate_kind_engine_values_and_states_from_documented_matrices():
  """X**p-prepared product states through each of the twelve kinds: the engine's expectation values and exported
  state against numbers computed from the closed-form matrices of tests/gate_docs.py alone (no oracle code)."""
  from tests import gate_docs as D
  rng = np.random.default_rng(120)
  code = {"I": (0, 0), "X": (1, 0), "Y": (1, 1), "Z": (0, 1)}
  for kind in range(1, 12):
    nq = O.gate_num_qubits(kind)
    for _ in range(2):
      t = float(rng.uniform(-1.5, 1.5))
      probes = [float(rng.uniform(0.1, 0.9)) for _ in range(nq)]
      bits = rng.integers(0, 2, size=(1, nq)).astype(np.int8)
      gates = [(O.GATE_XPOW, q, -1, -1, 0.0, probes[q]) for q in range(nq)]
      gates.append((kind, 0, 1 if nq == 2 else -1, -1, 0.0, t))
      strings = ["X", "Y", "Z"] if nq == 1 else ["XI", "IY", "ZZ", "XY", "YZ", "ZX", "YY"]
      ops = [[(1.0, sum(code[ch][0] << q for q, ch in enumerate(st)), sum(code[ch][1] << q for q, ch in enumerate(st)))]
             for st in strings]
      want = D.probe_values(kind, t, bits[0], probes, strings)[None, :]
      eng = _engine(nq, gates, 0, ops)
      got = eng.expectation(bits, np.zeros(0, np.float32)).cpu().numpy()
      np.testing.assert_allclose(got, want, atol=2e-6, err_msg=f"kind {kind}")
      # ... and the exported state, global phase included, equals documented G(t) . (x) X**p |x>
      prep = D.documented_matrix(O.GATE_XPOW, probes[0])
      if nq == 2:
        prep = np.kron(prep, D.documented_matrix(O.GATE_XPOW, probes[1]))
      idx = int("".join(str(int(b)) for b in bits[0]), 2)
      psi = D.documented_matrix(kind, t) @ prep[:, idx]
      state = eng.statevector(bits, np.zeros(0, np.float32)).cpu().numpy()[0]
      np.testing.assert_allclose(state, psi, atol=2e-6, err_msg=f"kind {kind} state")


def test_z_and_cz_power_hand_derived_closed_forms_on_the_engine():
  """H Z**t |0>: <X> = cos pi t, <Y> = sin pi t;  (H x H) CZ**t |00>: <X0 X1> = <X0> = (1 + cos pi t) / 2,
  <X0 Z1> = (1 - cos pi t) / 2, <Y0 Z1> = -sin(pi t) / 2 (derivation: tests/test_oracle_kat.py) -- config 3's own
  diagonal-gate conventions, parametrised through params so that the gradient d/dt is checked too."""
  for t in (0.3, -0.85, 1.7):
    params = np.array([t], np.float32)
    gates = [(O.GATE_HPOW, 0, -1, -1, 0.0, 1.0), (O.GATE_ZPOW, 0, -1, 0, 1.0, 0.0)]
    ops = [[(1.0, 1, 0)], [(1.0, 1, 1)], [(1.0, 0, 1)]]
    eng = _engine(1, gates, 1, ops)
    bits = np.zeros((1, 1), np.int8)
    vals, jac = eng.expectation_jacobian(bits, params)
    cp, sp = math.cos(math.pi * t), math.sin(math.pi * t)
    np.testing.assert_allclose(vals.cpu().numpy(), [[cp, sp, 0.0]], atol=2e-6)
    np.testing.assert_allclose(jac.cpu().numpy()[0, :, 0], [-math.pi * sp, math.pi * cp, 0.0], atol=2e-5)
    gates = [(O.GATE_HPOW, 0, -1, -1, 0.0, 1.0), (O.GATE_HPOW, 1, -1, -1, 0.0, 1.0), (O.GATE_CZPOW, 0, 1, 0, 1.0, 0.0)]
    ops = [[(1.0, 3, 0)], [(1.0, 1, 0)], [(1.0, 2, 0)], [(1.0, 1, 2)], [(1.0, 1, 3)], [(1.0, 0, 3)]]
    eng = _engine(2, gates, 1, ops)
    bits = np.zeros((1, 2), np.int8)
    vals, jac = eng.expectation_jacobian(bits, params)
    np.testing.assert_allclose(vals.cpu().numpy(), [[(1 + cp) / 2, (1 + cp) / 2, (1 + cp) / 2, (1 - cp) / 2, -sp / 2, 0.0]],
                               atol=2e-6)
    np.testing.assert_allclose(jac.cpu().numpy()[0, :, 0],
                               [-math.pi * sp / 2, -math.pi * sp / 2, -math.pi * sp / 2, math.pi * sp / 2, -math.pi * cp / 2, 0.0],
                               atol=2e-5)


def test_alternating_gradient_masks_swap_cached_backward_plans():
  """two inference paths alternating on ONE engine (the same total circuit with the data half frozen,
  then fully trainable) must not re-plan on every call: the engine keeps the backward plan of each mask it has seen
  and swaps it back in.  Results equal those of fresh engines, bit for bit, in any order of use."""
  n, layers = 13, 3
  rng = np.random.default_rng(31)
  gates, names = O.hea_gates(n, layers, "am")
  P = len(names)
  params = rng.uniform(-1, 1, P).astype(np.float32)
  ops = [O.xxz_chain_op(n)]
  bits = _random_bits(rng, 3, n)
  up = rng.normal(size=(3, 1)).astype(np.float32)
  first_half = {g[3] for g in gates[:len(gates) // 2] if g[3] >= 0}
  masks = [np.array([p not in first_half for p in range(P)]), None, rng.random(P) < 0.5]
  fresh, fresh_engine_bytes = [], 0
  for m in masks:
    e = _engine(n, gates, P, ops, adjoint_tile_qubits=10)
    e.set_gradient_mask(m)
    fresh.append(e.expectation_vjp(bits, params, up)[1].clone())
    fresh_engine_bytes = max(fresh_engine_bytes, e.allocated_bytes())
  eng = _engine(n, gates, P, ops, adjoint_tile_qubits=10)
  builds = []
  for k in (0, 1, 2, 0, 1, 2, 1, 0):
    eng.set_gradient_mask(masks[k])
    _, g = eng.expectation_vjp(bits, params, up)
    assert torch.equal(g, fresh[k]), k
    builds.append(eng.plan_builds())
  # the second visit of a mask plans nothing: one forward plan for the model, one backward plan per distinct mask,
  # whatever the order of use (the engine's own counter -- not wall-clock time)
  assert [b for _, b in builds] == [1, 2, 3, 3, 3, 3, 3, 3], builds
  assert {f for f, _ in builds} == {1}, builds
  # the cached plans' device copies are part of what the engine says it holds
  assert eng.allocated_bytes() > fresh_engine_bytes, (eng.allocated_bytes(), fresh_engine_bytes)
  # per-state rows are refused after a mask change until a VJP has run under the new mask
  eng.set_gradient_mask(masks[2])
  with pytest.raises(E.EngineError):
    eng.state_gradients(3)


@pytest.mark.parametrize("case", ["38 shards", "2 x 19 ring", "4 x 10 low", "Z and ZZ"])
def test_many_diagonal_terms_measured_in_the_wide_last_pass_at_19_qubits(case):
  """Round 5 regression: at 19 qubits the forward plan's measuring pass is the WIDE last pass (a tile of 2^13), and >= 32
  diagonal terms go through its Walsh-Hadamard transform.  The scheduler cut the terms' classes at the plan's tile (2^12)
  instead of the pass's: every such value was wrong (found while testing another option; tests/sanitize/plan_fuzz.cpp now
  checks the class of every term against the pass's own K)."""
  n = 19
  rng = np.random.default_rng(1900)
  gates, names = O.hea_gates(n, 3, "wht19")
  params = rng.uniform(-1, 1, len(names))
  bits = rng.integers(0, 2, size=(2, n)).astype(np.int8)
  zz = lambda qs: [(float(rng.normal()), 0, (1 << q) | (1 << ((q + 1) % n))) for q in qs]
  ops = {"38 shards": [[t] for t in zz(range(n)) + zz(range(n))],
         "2 x 19 ring": [zz(range(n)), zz(range(n))],
         "4 x 10 low": [zz(range(10)) for _ in range(4)],
         "Z and ZZ": [zz(range(n)), [(float(rng.normal()), 0, 1 << q) for q in range(n)]]}[case]
  want = O.expectation(n, gates, params, bits, ops)   # (values only: a Jacobian at 19 qubits costs minutes)
  eng = _engine(n, gates, len(names), ops)
  forward = eng.describe_schedule().split("adjoint")[0]
  measuring = [line for line in forward.splitlines() if "meas_terms=" in line and "meas_terms=0" not in line]
  assert measuring and all("K=13" in line for line in measuring), measuring   # (the case the test is for)
  got = eng.expectation(bits, params).cpu().numpy()
  np.testing.assert_allclose(got, want, atol=5e-5 * max(1.0, _op_norm(ops).max()), rtol=0)
