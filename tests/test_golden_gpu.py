"""The HIP engine against the committed golden vectors and, at BASELINE.json's full
sizes, against size-independent properties (GPU only).

Tolerances (fp32 engine vs complex128 golden): values 1e-5 * sum|c_k| (n <= 12),
gradients 1e-4 * max(1, |grad|_inf); full-size properties as stated per test.
"""
import numpy as np
import pytest
import torch

import bench
from qhbmlib_amd import _engine as E
from tests import golden_util as G

pytestmark = pytest.mark.gpu


def _norms(ops):
  return np.array([max(1.0, sum(abs(c) for c, _, _ in op)) for op in ops])


def _engine(n, gates, n_params, ops, **opts):
  eng = E.Engine(0)
  for k, v in opts.items():
    eng.set_option(k, v)
  eng.set_circuit(n, gates, n_params)
  eng.set_observables(ops)
  return eng


@pytest.mark.parametrize("name", G.hea_files() + ["all_kinds_n5.npz"])
def test_engine_reproduces_golden(name):
  g = G.load(name)
  n, gates, ops = int(g["n"]), G.gates_of(g["gates"]), G.ops_of(g["ops"])
  eng = _engine(n, gates, len(g["params"]), ops)
  vals, jac = eng.expectation_jacobian(g["bits"], g["params"])
  assert (np.abs(vals.cpu().numpy() - g["values"]) <= 1e-5 * _norms(ops)[None, :]).all()
  np.testing.assert_allclose(jac.cpu().numpy(), g["jacobian"],
                             atol=1e-4 * max(1.0, np.abs(g["jacobian"]).max()), rtol=0)


@pytest.mark.parametrize("name", G.hea_files())
@pytest.mark.parametrize("tag", ["bernoulli", "kobe2"])
def test_engine_modular_hamiltonian_golden(name, tag):
  g = G.load(name)
  n, gates, shards = int(g["n"]), G.gates_of(g["total_gates"]), G.ops_of(g[f"{tag}_shards"])
  eng = _engine(n, gates, len(g["total_params"]), shards)
  vals, jac = eng.expectation_jacobian(g["bits"], g["total_params"])
  np.testing.assert_allclose(vals.cpu().numpy(), g[f"{tag}_values"], atol=1e-5, rtol=0)
  np.testing.assert_allclose(jac.cpu().numpy(), g[f"{tag}_jacobian"],
                             atol=1e-4 * max(1.0, np.abs(g[f"{tag}_jacobian"]).max()), rtol=0)


def test_engine_vqt_fixture():
  """BASELINE config 1 on a fixed multiset: loss and d/dphi through values + one VJP."""
  g = G.load("vqt_c1.npz")
  n, gates, target = int(g["n"]), G.gates_of(g["gates"]), G.ops_of(g["target"])
  samples = g["samples"]
  uniq, counts = np.unique(samples, axis=0, return_counts=True)
  eng = _engine(n, gates, len(g["params"]), target)
  w = counts / counts.sum()
  vals, grad = eng.expectation_vjp(uniq, g["params"], float(g["beta"]) * w[:, None])
  energies = (1 - 2 * uniq.astype(np.float64)) @ g["thetas"]
  loss = float(w @ (float(g["beta"]) * vals.cpu().numpy()[:, 0] - energies) - g["log_partition"])
  np.testing.assert_allclose(loss, float(g["loss"]), atol=2e-5)
  np.testing.assert_allclose(grad.cpu().numpy(), g["dparams"], atol=1e-4)


# ---- BASELINE config 2 at full size: 12 qubits, depth-8 HEA, TFIM, 1024 states ----------------
def test_c2_full_size_against_c_oracle_sample_and_properties():
  n, layers, states = 12, 8, 1024
  gates, n_params = bench.hea_gates(n, layers)
  op = bench.tfim_op(n)
  params = np.random.default_rng(1234).uniform(-1, 1, n_params).astype(np.float32)
  bits = bench.distinct_bitstrings(n, states, 4321)
  identity = [(1.0, 0, 0)]
  eng = _engine(n, gates, n_params, [op, identity, [(2.0 * c, x, z) for c, x, z in op]])
  vals = eng.expectation(bits, params).cpu().numpy()
  np.testing.assert_allclose(vals[:, 1], 1.0, atol=2e-5)               # norm preserved
  np.testing.assert_allclose(vals[:, 2], 2.0 * vals[:, 0], atol=5e-5)  # linearity
  from oracle import qhbm_cpu as C
  pick = np.arange(0, states, 64)
  want = C.expectation(n, gates, params, bits[pick], [op])
  np.testing.assert_allclose(vals[pick, 0], want[:, 0], atol=1e-5 * 24 * 4)
  # duplicates and row order (qnn_test.py:437-442)
  dup = np.concatenate([bits[:8], bits[:8][::-1]])
  got = eng.expectation(dup, params).cpu().numpy()
  # equal up to the order of the fp32 atomic partial sums (tile/wave reductions are unordered)
  np.testing.assert_allclose(got[:8], got[8:][::-1], atol=2e-6 * 24)


# ---- BASELINE config 3 shape at full qubit count: 20 qubits, depth 16, XXZ ----------------------
def test_c3_full_size_properties():
  n, layers, states = 20, 16, 16
  gates, n_params = bench.hea_gates(n, layers)
  op = bench.xxz_op(n)
  rng = np.random.default_rng(20)
  params = rng.uniform(-1, 1, n_params).astype(np.float32)
  bits = bench.distinct_bitstrings(n, states, 7)
  identity = [(1.0, 0, 0)]
  eng = _engine(n, gates, n_params, [op, identity])
  vals = eng.expectation(bits, params).cpu().numpy()
  np.testing.assert_allclose(vals[:, 1], 1.0, atol=5e-5)   # unitarity over 944 gates in fp32
  assert np.abs(vals[:, 0]).max() < 57                     # |<H>| <= sum |c_k|
  # adjoint gradient == two-term shift rule, evaluated by the engine's own forward (each HEA
  # parameter drives exactly one gate): d<H>/dp = (pi/2) [<H>(p + 1/2) - <H>(p - 1/2)]
  up = np.zeros((states, 2), np.float32)
  up[:, 0] = 1.0 / states
  _, grad = eng.expectation_vjp(bits, params, up)
  grad = grad.cpu().numpy()
  for p in (0, 17, n_params // 2, n_params - 1):
    hi, lo = params.copy(), params.copy()
    hi[p] += 0.5
    lo[p] -= 0.5
    fd = (np.pi / 2) * (eng.expectation(bits, hi).cpu().numpy()[:, 0].mean() -
                        eng.expectation(bits, lo).cpu().numpy()[:, 0].mean())
    np.testing.assert_allclose(grad[p], fd, atol=2e-4)
  # U followed by U^-1 returns the basis state: <Z_q> = (-1)^{x_q} exactly
  inv = [(k, q0, q1, p, -s, -o) for (k, q0, q1, p, s, o) in reversed(gates)]
  z_ops = [[(1.0, 0, 1 << q)] for q in range(n)]
  eng2 = _engine(n, gates + inv, n_params, z_ops)
  z = eng2.expectation(bits[:4], params).cpu().numpy()
  np.testing.assert_allclose(z, 1.0 - 2.0 * bits[:4], atol=2e-4)


def test_tile_geometries_agree_at_n20():
  """Different LDS tilings / round widths are different programs for the same circuit."""
  n, layers, states = 20, 4, 4
  gates, n_params = bench.hea_gates(n, layers)
  params = np.random.default_rng(3).uniform(-1, 1, n_params).astype(np.float32)
  bits = bench.distinct_bitstrings(n, states, 11)
  up = np.full((states, 1), 0.25, np.float32)
  ref_vals = ref_grad = None
  for tile, rnd, adj in ((13, 4, 12), (12, 4, 11), (14, 4, 13), (11, 4, 10)):
    eng = _engine(n, gates, n_params, [bench.xxz_op(n)], tile_qubits=tile, round_qubits=rnd,
                  adjoint_tile_qubits=adj)
    vals, grad = eng.expectation_vjp(bits, params, up)
    vals, grad = vals.cpu().numpy(), grad.cpu().numpy()
    if ref_vals is None:
      ref_vals, ref_grad = vals, grad
    np.testing.assert_allclose(vals, ref_vals, atol=5e-5)
    np.testing.assert_allclose(grad, ref_grad, atol=5e-5)
