"""The block-grouped Pauli-sum kernels (csrc/observable.hip) against the oracle and against the gather kernel.

lambda = O psi and <psi|O_t|psi> for operators with many X-masks: partner blocks staged in LDS, 64 straight-line
sign / half / imaginary variants behind a jump table.  Every variant bit, every row permutation, every mode
(lambda, lambda + value, values, values of several observables) and both XCD maps are exercised against the numpy
complex128 oracle (reference pattern: tests/inference/qnn_test.py:183-264 -- simulate, compare, assert).

Tolerances: expectations 5e-5 * sum|c_k|, gradients 1e-4 * max(1, |grad|_inf).
"""
import numpy as np
import pytest
import torch

from oracle import qhbm_oracle as O
from qhbmlib_amd import _engine as E

pytestmark = pytest.mark.gpu


def _engine(n, gates, n_params, ops, **options):
  eng = E.Engine(0)
  for k, v in options.items():
    eng.set_option(k, v)
  eng.set_circuit(n, gates, n_params)
  eng.set_observables(ops)
  return eng


def _random_ops(rng, n, n_ops, terms_per_op, p_identity=0.6):
  """Random Pauli sums: X / Y / Z on every qubit with equal weight, so that odd Y counts (imaginary weights), odd
  index-bit-0 flips and Z bits on every slot position all occur."""
  ops = []
  for _ in range(n_ops):
    op = []
    while len(op) < terms_per_op:
      kinds = rng.choice(4, size=n, p=[p_identity] + [(1 - p_identity) / 3] * 3)
      if not kinds.any():
        continue
      x = sum(1 << q for q in range(n) if kinds[q] in (1, 2))
      z = sum(1 << q for q in range(n) if kinds[q] in (2, 3))
      op.append((float(rng.normal()), x, z))
    ops.append(op)
  return ops


def _check(eng, n, gates, params, bits, ops, up):
  if n >= 16:   # the fp32 C oracle (seconds) where the numpy one needs a minute per test; tolerances unchanged
    from oracle import qhbm_cpu as C
    want_vals, want_grad = C.expectation_vjp(n, gates, params, bits, ops, up)
  else:
    want_vals, want_jac = O.expectation_jacobian(n, gates, params, bits, ops)
    want_grad = np.einsum("bt,btp->p", up, want_jac)
  norm = np.array([sum(abs(c) for c, _, _ in op) for op in ops])
  tol_v = 5e-5 * np.maximum(norm, 1.0)[None, :]
  tol_g = 1e-4 * max(1.0, np.abs(want_grad).max())
  # forward only
  got = eng.expectation(bits, params).cpu().numpy()
  assert (np.abs(got - want_vals) <= tol_v).all(), np.abs(got - want_vals).max()
  # values + adjoint VJP in one call
  vals, grad = eng.expectation_vjp(bits, params, up)
  assert (np.abs(vals.cpu().numpy() - want_vals) <= tol_v).all(), np.abs(vals.cpu().numpy() - want_vals).max()
  np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=tol_g, rtol=0)
  # the autograd pair: retained forward, backward later
  vals2 = eng.expectation(bits, params, retain=True)
  assert (np.abs(vals2.cpu().numpy() - want_vals) <= tol_v).all()
  if eng.retained is not None:
    grad2 = eng.expectation_vjp_retained(bits, params, up)
    np.testing.assert_allclose(grad2.cpu().numpy(), want_grad, atol=tol_g, rtol=0)
  return vals.cpu().numpy(), grad.cpu().numpy()


@pytest.mark.parametrize("block_bits", [13, 12])   # the shipped shape, and two independent workgroups per CU (option-only)
@pytest.mark.parametrize("n", [13, 15])
@pytest.mark.parametrize("n_ops", [1, 3])
def test_random_pauli_sums_block_kernel_against_oracle(n, n_ops, block_bits):
  rng = np.random.default_rng(100 * n + n_ops)
  gates, names = O.hea_gates(n, 2, "blk")
  params = rng.uniform(-1, 1, len(names))
  ops = _random_ops(rng, n, n_ops, 40)
  bits = rng.integers(0, 2, size=(5, n)).astype(np.int8)
  up = rng.normal(size=(5, n_ops))
  eng = _engine(n, gates, len(names), ops, tile_qubits=10, adjoint_tile_qubits=10, observable_kernel=1,
                multi_observable_values=1, observable_block_bits=block_bits)
  v_blk, g_blk = _check(eng, n, gates, params, bits, ops, up)
  # the gather kernel on the same inputs (the values of several observables are then measured in the passes)
  ref = _engine(n, gates, len(names), ops, tile_qubits=10, adjoint_tile_qubits=10, observable_kernel=0)
  v_ref, g_ref = ref.expectation_vjp(bits, params, up)
  np.testing.assert_allclose(v_blk, v_ref.cpu().numpy(), atol=2e-5 * 40, rtol=0)
  np.testing.assert_allclose(g_blk, g_ref.cpu().numpy(), atol=1e-4 * max(1.0, float(np.abs(g_blk).max())), rtol=0)


@pytest.mark.parametrize("block_bits", [13, 12, 113])   # 113: blocks of 2^13, the halves split the rows (observable_split_rows)
def test_every_sign_half_and_imaginary_variant_one_term_at_a_time(block_bits):
  """One observable per (slot Z bits, odd x, Y parity) combination at 13 qubits, i.e. one term per jump-table chunk:
  slot bits of the block layout are index bits 0, 10, 11, 12 = qubits 12, 2, 1, 0 (blocks of 2^12: bit 12 is a block bit)."""
  n = 13
  rng = np.random.default_rng(7)
  gates, names = O.hea_gates(n, 2, "var")
  params = rng.uniform(-1, 1, len(names))
  bits = rng.integers(0, 2, size=(3, n)).astype(np.int8)
  slot_qubits = [12, 2, 1, 0]   # index bit 0, 10, 11, 12 (qubit q <-> index bit n - 1 - q)
  ops = []
  for zs in range(16):
    for odd in (0, 1):
      for imag in (0, 1):
        x = z = 0
        for b, q in enumerate(slot_qubits):
          if zs >> b & 1:
            z |= 1 << q
        if odd:
          x |= 1 << 12           # X (or Y) on the qubit of index bit 0
        x |= 1 << 6              # a thread-bit flip, so that the partner is another lane's pair
        if imag:                 # an odd number of Y factors: make qubit 6 a Y
          z |= 1 << 6
        # keep the Y parity as requested even when the slot Z bits put a Y on qubit 12
        ny = bin(x & z).count("1")
        if (ny & 1) != imag:
          z ^= 1 << 5
          x |= 1 << 5            # X -> Y (or the reverse) on another thread bit flips the parity
        assert (bin(x & z).count("1") & 1) == imag
        ops.append([(float(rng.uniform(0.5, 1.5)), x, z)])
  eng = _engine(n, gates, len(names), ops, tile_qubits=10, adjoint_tile_qubits=10, observable_kernel=1,
                multi_observable_values=1, observable_block_bits=min(block_bits, 13), observable_split_rows=int(block_bits == 113))
  up = rng.normal(size=(3, len(ops)))
  _check(eng, n, gates, params, bits, ops, up)


@pytest.mark.parametrize("block_bits", [13, 12, 113])
@pytest.mark.parametrize("xcd", [0, 1])
def test_many_masks_at_19_qubits_values_lambda_and_both_xcd_maps(xcd, block_bits):
  """64 (128) blocks per state: partner blocks of other workgroups, the pair-halving of the value modes, the XCD maps."""
  n = 19
  rng = np.random.default_rng(19 + xcd)
  gates, names = O.hea_gates(n, 3, "m19")
  params = rng.uniform(-1, 1, len(names))
  ops = _random_ops(rng, n, 1, 96, p_identity=0.75)
  bits = rng.integers(0, 2, size=(9, n)).astype(np.int8)   # 9: one group of eight states and a remainder
  up = rng.normal(size=(9, 1))
  eng = _engine(n, gates, len(names), ops, observable_kernel=1, observable_xcd_states=xcd, observable_block_bits=min(block_bits, 13),
                observable_split_rows=int(block_bits == 113))
  _check(eng, n, gates, params, bits, ops, up)


@pytest.mark.parametrize("n_ops", [1, 3])
def test_two_level_sweep_far_windows_of_the_gather_kernel_against_the_oracle(n_ops):
  """Round 5: masks that flip only far index bits go to FAR launches of apply_observable_kernel (virtual index space
  with the window swapped into bits 4..10, lambda and value partials accumulated onto the first launch's).  Forced at
  18 qubits (window = bits 11..17): single flips on every qubit, ZZ pairs, Y terms on far bits (imaginary weights),
  masks that mix window bits with bits 0..3 (taken) and with bits 4..10 (left to the first launch)."""
  n = 18
  rng = np.random.default_rng(180 + n_ops)
  gates, names = O.hea_gates(n, 2, "far")
  params = rng.uniform(-1, 1, len(names))
  def op():
    terms = [(float(rng.normal()), 1 << q, 0) for q in range(n)]                                  # X_q
    terms += [(float(rng.normal()), 0, (1 << q) | (1 << ((q + 1) % n))) for q in range(n)]        # Z_q Z_q+1
    terms += [(float(rng.normal()), 1 << q, 1 << q) for q in (0, 3, 11, 14, 17)]                  # Y_q
    terms += [(float(rng.normal()), (1 << 12) | (1 << 16), 1 << 12), (float(rng.normal()), (1 << 13) | (1 << 2), (1 << 2) | (1 << 9)),
              (float(rng.normal()), (1 << 15) | (1 << 6), 1 << 15), (float(rng.normal()), (1 << 17) | (1 << 11) | 1, (1 << 17) | 1),
              (float(rng.normal()), (1 << 14) | (1 << 10), 0)]
    return terms
  ops = [op() for _ in range(n_ops)]
  bits = rng.integers(0, 2, size=(3, n)).astype(np.int8)
  up = rng.normal(size=(3, n_ops))
  eng = _engine(n, gates, len(names), ops, observable_kernel=0, observable_far_windows=1)
  vals, grad = _check(eng, n, gates, params, bits, ops, up)
  ref = _engine(n, gates, len(names), ops, observable_kernel=0, observable_far_windows=0)
  rv, rg = ref.expectation_vjp(bits, params, up)
  np.testing.assert_allclose(vals, rv.cpu().numpy(), atol=2e-5, rtol=0)
  np.testing.assert_allclose(grad, rg.cpu().numpy(), atol=2e-5 * max(1.0, np.abs(grad).max()), rtol=0)


@pytest.mark.parametrize("kernel", [-1, 1])
def test_three_observables_xxz_split_at_16_qubits_match_the_single_sum(kernel):
  """XXZ split into its XX, YY and ZZ sums (the reference's normal usage: several operators per call,
  tests/inference/qnn_test.py:187-190): the three values add up to the one-observable value, gradients agree."""
  n = 16
  rng = np.random.default_rng(16)
  gates, names = O.hea_gates(n, 3, "x3")
  params = rng.uniform(-1, 1, len(names))
  xx, yy, zz = [], [], []
  for i in range(n - 1):
    m = (1 << i) | (1 << (i + 1))
    xx.append((1.0, m, 0))
    yy.append((1.0, m, m))
    zz.append((0.5, 0, m))
  bits = rng.integers(0, 2, size=(6, n)).astype(np.int8)
  up3 = rng.normal(size=(6, 3))
  eng3 = _engine(n, gates, len(names), [xx, yy, zz], observable_kernel=kernel)   # (-1: values by the block kernel, lambda by the gather kernel)
  assert eng3.num_passes()[0] > 1
  v3, _ = _check(eng3, n, gates, params, bits, [xx, yy, zz], up3)
  eng1 = _engine(n, gates, len(names), [xx + yy + zz])
  v1 = eng1.expectation(bits, params).cpu().numpy()
  np.testing.assert_allclose(v3.sum(1, keepdims=True), v1, atol=5e-5 * (2 * (n - 1) + 0.5 * (n - 1)), rtol=0)


@pytest.mark.parametrize("n_ops,n", [(2, 12), (3, 14), (4, 15)])
def test_gather_kernel_with_an_accumulator_per_observable_matches_the_oracle(n_ops, n):
  """Round 5: two to four observables through apply_observable_kernel<A, OBS_GATHER_MULTI> -- the weighted lambda AND
  every <psi|O_t|psi> from one launch ("gather_multi_values" = 1; not the default: measured slower than two launches at
  config 3's size, profiles/r05_xxz3_gather_multi_ab.txt) -- on random Pauli sums with real and imaginary weights (odd Y counts), masks
  inside and outside the kernel's block, shared masks between observables; values-only calls, VJP calls, the retained
  pair, chunked batches; against the numpy oracle and bit-identical from call to call."""
  rng = np.random.default_rng(100 * n_ops + n)
  gates, names = O.hea_gates(n, 2, "gm")
  params = rng.uniform(-1, 1, len(names))
  ops = _random_ops(rng, n, n_ops, 7, p_identity=0.6)
  ops[1] = ops[1] + [(0.5, x, z ^ 1) for _, x, z in ops[0][:3]]   # masks shared between observables
  bits = rng.integers(0, 2, size=(9, n)).astype(np.int8)
  up = rng.normal(size=(9, n_ops))
  eng = _engine(n, gates, len(names), ops, gather_multi_values=1, tile_qubits=10, adjoint_tile_qubits=10)
  assert "an accumulator per observable" in eng.describe_schedule()
  v, g = _check(eng, n, gates, params, bits, ops, up)
  v2, g2 = eng.expectation_vjp(bits, params, up)
  np.testing.assert_array_equal(v, v2.cpu().numpy())                # bit-identical from call to call
  np.testing.assert_array_equal(g, g2.cpu().numpy())
  fwd = eng.expectation(bits, params).cpu().numpy()
  np.testing.assert_array_equal(fwd, v2.cpu().numpy())             # values-only launch: the same accumulators
  eng.expectation(bits, params, retain=True)
  if eng.retained is not None:
    g_r = eng.expectation_vjp_retained(bits, params, up).cpu().numpy()
    np.testing.assert_allclose(g_r, g2.cpu().numpy(), atol=2e-6 * max(1.0, float(np.abs(g_r).max())), rtol=0)
  chunked = _engine(n, gates, len(names), ops, gather_multi_values=1, tile_qubits=10, adjoint_tile_qubits=10, chunk_states=4)
  v3, g3 = chunked.expectation_vjp(bits, params, up)
  assert torch.equal(v2, v3) and torch.equal(g2, g3)
  # the default path (values from the block kernel's launch, where the state has a block) agrees
  if n >= 13:
    old = _engine(n, gates, len(names), ops, multi_observable_values=1, tile_qubits=10, adjoint_tile_qubits=10)
    assert "an accumulator per observable" not in old.describe_schedule()
    v4, g4 = old.expectation_vjp(bits, params, up)
    np.testing.assert_allclose(v4.cpu().numpy(), v2.cpu().numpy(), atol=2e-5 * max(sum(abs(c) for c, _, _ in op) for op in ops), rtol=0)
    np.testing.assert_allclose(g4.cpu().numpy(), g2.cpu().numpy(), atol=1e-4 * max(1.0, float(g2.abs().max())), rtol=0)


def test_parameter_shift_takes_its_values_from_the_block_kernel():
  n = 14
  rng = np.random.default_rng(14)
  gates, names = O.hea_gates(n, 1, "ps")
  params = rng.uniform(-1, 1, len(names))
  ops = _random_ops(rng, n, 1, 24, p_identity=0.7)
  bits = rng.integers(0, 2, size=(2, n)).astype(np.int8)
  up = rng.normal(size=(2, 1))
  eng = _engine(n, gates, len(names), ops, tile_qubits=10, observable_kernel=1)
  vals, g_shift = eng.expectation_vjp(bits, params, up, method=E.GRAD_PARAMETER_SHIFT)
  _, g_adj = eng.expectation_vjp(bits, params, up)
  want_vals, want_jac = O.expectation_jacobian(n, gates, params, bits, ops)
  want = np.einsum("bt,btp->p", up, want_jac)
  tol = 1e-4 * max(1.0, np.abs(want).max())
  np.testing.assert_allclose(g_shift.cpu().numpy(), want, atol=2 * tol, rtol=0)
  np.testing.assert_allclose(g_adj.cpu().numpy(), want, atol=tol, rtol=0)
  np.testing.assert_allclose(vals.cpu().numpy(), want_vals, atol=5e-5 * sum(abs(c) for c, _, _ in ops[0]), rtol=0)


def test_values_are_bit_reproducible_and_independent_of_chunking():
  n = 15
  rng = np.random.default_rng(15)
  gates, names = O.hea_gates(n, 2, "rep")
  params = rng.uniform(-1, 1, len(names))
  ops = _random_ops(rng, n, 2, 30)
  bits = rng.integers(0, 2, size=(12, n)).astype(np.int8)
  up = rng.normal(size=(12, 2))
  eng = _engine(n, gates, len(names), ops, observable_kernel=1, multi_observable_values=1, tile_qubits=10,
                adjoint_tile_qubits=10)
  v0, g0 = eng.expectation_vjp(bits, params, up)
  v1, g1 = eng.expectation_vjp(bits, params, up)
  assert torch.equal(v0, v1) and torch.equal(g0, g1)
  rows0 = eng.state_gradients(12).clone()
  chunked = _engine(n, gates, len(names), ops, observable_kernel=1, multi_observable_values=1, tile_qubits=10,
                    adjoint_tile_qubits=10, chunk_states=5)
  v2, _ = chunked.expectation_vjp(bits, params, up)
  assert torch.equal(v0, v2)
  assert torch.equal(rows0, chunked.state_gradients(12))
