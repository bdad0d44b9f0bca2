"""The sizes where the planner changes its mind, in the DRIVER-run suite (round 5's review, item 1).

The only wrong-value bug of the project lived on a default path between two fixtures (19 qubits, HISTORY round 5); 15..19
are covered by tests/test_default_plans_gpu.py.  Here: 21, 22, 23 qubits (the forward sweep switches to tiles of 2^13 at 22,
lambda = O psi changes its XCD map at 23 / 64 MiB per state) with values AND the VJP under default options and
`wide_last_pass=0`; 25, 26, 27 qubits (one state, <= 24 terms, depth 2); the `Hamiltonian` branch THROUGH the host mirror
(the user-facing form of the path that was wrong) at 13, 17, 19, 22 qubits; the reference's own bit order
(`tfq_compat_bit_order=True`, /root/reference/qhbmlib/models/circuit.py:59-62,131-134) at 20 and 24 qubits; the other entry
points (statevector, retained pair, shift against adjoint, sampling) at 21..23.

Checker: oracle/qhbm_cpu.c (fp32, gate by gate; threads inside a state at these sizes), itself held to the numpy complex128
oracle in tests/test_oracle_c.py.  Reference pattern: tests/inference/qnn_test.py:183-264,266-369 (simulate, compare).
Tolerances: values 5e-5 * max(1, sum|c_k|); gradients 3e-4 * max(1, |grad|_inf) (both sides fp32 here).
"""
import numpy as np
import pytest
import torch

from oracle import qhbm_cpu as C
from oracle import qhbm_oracle as O
from qhbmlib_amd import _engine as E
from qhbmlib_amd import inference, ir, models
from tests.test_host_api import hea_circuit

pytestmark = pytest.mark.gpu


def _engine(n, gates, n_params, ops, **options):
  eng = E.Engine(0)
  for k, v in options.items():
    eng.set_option(k, v)
  eng.set_circuit(n, gates, n_params)
  eng.set_observables(ops)
  return eng


def _norm(ops):
  return np.maximum(np.array([sum(abs(c) for c, _, _ in op) for op in ops]), 1.0)


def _set(param, values):
  with torch.no_grad():
    param.copy_(torch.as_tensor(np.asarray(values), dtype=torch.float32))


# ---- (a) 21..23 qubits: values + VJP, three observable layouts, default options and wide_last_pass = 0 -----------------
@pytest.mark.parametrize("n", [21, 22, 23])
def test_values_and_vjp_under_default_plans_at_21_to_23_qubits(n):
  rng = np.random.default_rng(2100 + n)
  gates, names = O.hea_gates(n, 2 + n % 2, "ps")
  P = len(names)
  params = rng.uniform(-1, 1, P).astype(np.float32)
  chain = [(float(rng.normal()), 0, (1 << q) | (1 << ((q + 1) % n))) for q in range(n)]
  chain += [(float(rng.normal()), 0, 1 << q) for q in range(n)]
  scattered = [(float(rng.normal()), 0, int(sum(1 << int(q) for q in rng.choice(n, size=int(rng.integers(1, 4)), replace=False))))
               for _ in range(12)]
  flips = [(float(rng.normal()), 1 << int(q), 0) for q in rng.choice(n, size=3, replace=False)]
  flips.append(O.pauli_term(float(rng.normal()), [(0, "X"), (n - 1, "Y"), (n // 2, "Z")]))
  layouts = {"shards": [[t] for t in chain + scattered] + [flips],                       # >= 32 diagonal observables
             "few": [chain[0::3] + flips[:1], chain[1::3] + scattered[:6], chain[2::3] + scattered[6:] + flips[3:]],
             "one": [chain + scattered + flips]}
  bits = rng.integers(0, 2, size=(2, n)).astype(np.int8)
  for name, ops in layouts.items():
    up = rng.normal(size=(2, len(ops))).astype(np.float32)
    want, want_grad = C.expectation_vjp(n, gates, params, bits, ops, up)
    norm = _norm(ops)
    for opts in ({}, {"wide_last_pass": 0}):
      eng = _engine(n, gates, P, ops, **opts)
      got = eng.expectation(bits, params).cpu().numpy()
      assert (np.abs(got - want) / norm[None, :]).max() <= 5e-5, (name, opts, "forward only")
      vals, grad = eng.expectation_vjp(bits, params, up)
      assert (np.abs(vals.cpu().numpy() - want) / norm[None, :]).max() <= 5e-5, (name, opts, "values of the VJP call")
      np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=3e-4 * max(1.0, float(np.abs(want_grad).max())),
                                 rtol=0, err_msg=f"{name} {opts}")


# ---- (a) 25..27 qubits: one state, <= 24 terms, depth 2 ------------------------------------------------------------------
@pytest.mark.parametrize("n", [25, 26, 27])
def test_values_and_vjp_of_one_state_at_25_to_27_qubits(n):
  rng = np.random.default_rng(2500 + n)
  gates, names = O.hea_gates(n, 2, "big")
  P = len(names)
  params = rng.uniform(-1, 1, P).astype(np.float32)
  qs = rng.choice(n, size=10, replace=False)
  op_a = [(float(rng.normal()), 0, (1 << int(q)) | (1 << int((q + 1) % n))) for q in qs] + \
         [(float(rng.normal()), 1 << int(q), 0) for q in qs[:6]]
  op_b = [O.pauli_term(float(rng.normal()), [(int(qs[0]), "X"), (int(qs[1]), "X")]),
          O.pauli_term(float(rng.normal()), [(int(qs[0]), "Y"), (int(qs[1]), "Y")]),
          O.pauli_term(float(rng.normal()), [(0, "Y"), (n - 1, "Z")]),
          O.pauli_term(float(rng.normal()), [(n - 1, "X"), (3, "Z"), (n // 2, "Z")])]
  ops = [op_a, op_b]                                                                       # 20 terms
  bits = rng.integers(0, 2, size=(1, n)).astype(np.int8)
  up = rng.normal(size=(1, 2)).astype(np.float32)
  want, want_grad = C.expectation_vjp(n, gates, params, bits, ops, up)
  eng = _engine(n, gates, P, ops)
  got = eng.expectation(bits, params).cpu().numpy()
  assert (np.abs(got - want) / _norm(ops)[None, :]).max() <= 5e-5
  vals, grad = eng.expectation_vjp(bits, params, up)
  assert (np.abs(vals.cpu().numpy() - want) / _norm(ops)[None, :]).max() <= 5e-5
  np.testing.assert_allclose(grad.cpu().numpy(), want_grad, atol=3e-4 * max(1.0, float(np.abs(want_grad).max())), rtol=0)


# ---- (b) the Hamiltonian branch through the mirror: qnn.expectation(bits, Hamiltonian(KOBE-2, V)) ------------------------
@pytest.mark.parametrize("n", [13, 17, 19, 22])
def test_modular_hamiltonian_through_the_mirror_values_and_gradients(n):
  """/root/reference/tests/inference/qnn_test.py:266-369: <x|U^dag (V diag(E_theta) V^dag) U|x> as n + n(n-1)/2 Z-string
  shards measured behind U V^dag and combined with theta; gradients wrt theta, phi_V and phi_U by torch.autograd through
  the engine's adjoint sweep.  At these sizes the shards (91 .. 253 >= 32) take the Walsh-Hadamard measurement."""
  qubits = ir.GridQubit.rect(1, n)
  rng = np.random.default_rng(300 + n)
  energy_h = models.KOBE(list(range(n)), 2)
  circuit_h = models.DirectQuantumCircuit(hea_circuit(qubits, 1, "h"), tfq_compat_bit_order=False)
  model = models.DirectQuantumCircuit(hea_circuit(qubits, 2, "m"), tfq_compat_bit_order=False)
  h_vals = rng.uniform(-1, 1, len(circuit_h.symbol_names)).astype(np.float32)
  m_vals = rng.uniform(-1, 1, len(model.symbol_names)).astype(np.float32)
  thetas = rng.uniform(-1, 1, energy_h.post_process[0].kernel.numel()).astype(np.float32)
  _set(circuit_h.trainable_variables[0], h_vals)
  _set(model.trainable_variables[0], m_vals)
  _set(energy_h.post_process[0].kernel, thetas)
  bits = rng.integers(0, 2, size=(3, n)).astype(np.int8)
  bits = np.concatenate([bits, bits[1:2]])                                                # a duplicate row (qnn_test.py:437-442)
  weights = rng.normal(size=(4,)).astype(np.float32)
  qnn = inference.AnalyticQuantumInference(model)
  variables = energy_h.trainable_variables + circuit_h.trainable_variables + model.trainable_variables
  out = qnn.expectation(torch.from_numpy(bits), models.Hamiltonian(energy_h, circuit_h))
  assert out.shape == (4, 1)
  loss = (out[:, 0] * torch.as_tensor(weights, device=out.device)).sum()
  g_theta, g_h, g_m = torch.autograd.grad(loss, variables)
  # oracle: parameters = [model..., hamiltonian...], circuit = U then V^dag
  all_names = model.symbol_names + circuit_h.symbol_names
  total = model.pqc.flat_gates(qubits, all_names) + O.inverse_gates(circuit_h.pqc.flat_gates(qubits, all_names))
  params = np.concatenate([m_vals, h_vals])
  shards = O.kobe_shards(n, 2)
  shard_vals = C.expectation(n, total, params, bits, shards)                              # [4, T_s]
  np.testing.assert_allclose(out.detach().cpu().numpy()[:, 0], shard_vals @ thetas,
                             atol=5e-5 * max(1.0, float(np.abs(thetas).sum())))
  np.testing.assert_allclose(g_theta.cpu().numpy().reshape(-1), weights @ shard_vals, atol=1e-4)
  combined = [[(float(t) * c, x, z) for t, op in zip(thetas, shards) for c, x, z in op]]
  _, want_phi = C.expectation_vjp(n, total, params, bits, combined, weights[:, None])
  scale = max(1.0, float(np.abs(want_phi).max()))
  np.testing.assert_allclose(g_m.cpu().numpy().reshape(-1), want_phi[: len(m_vals)], atol=3e-4 * scale)
  np.testing.assert_allclose(g_h.cpu().numpy().reshape(-1), want_phi[len(m_vals):], atol=3e-4 * scale)


# ---- (c) the reference's actual bit order from 11 qubits on, at the BASELINE sizes ------------------------------------------
@pytest.mark.parametrize("n,layers", [(20, 2), (24, 1)])
def test_tfq_compat_bit_order_through_the_mirror_at_baseline_sizes(n, layers):
  """/root/reference/qhbmlib/models/circuit.py:59-62,131-134: the injector's symbols are sorted as STRINGS, so bitstring
  column j drives qubit perm[j] (`bit_10` < `bit_2`).  With the flag the mirror must equal the oracle run on the permuted
  bitstrings -- and must DIFFER from the default order on the same inputs."""
  qubits = ir.GridQubit.rect(1, n)
  rng = np.random.default_rng(40 + n)
  bits = rng.integers(0, 2, size=(3, n)).astype(np.int8)
  perm = O.tfq_bit_permutation(n)
  assert perm != list(range(n))
  assert (O.apply_bit_order(bits, True) != bits).any(axis=1).all()                       # (no row is a fixed point)
  ops = [ir.PauliSum.from_pauli_strings([ir.PZ(q) * (i + 1.0) for i, q in enumerate(qubits)]),
         ir.PauliSum.from_pauli_strings([ir.PX(qubits[i]) * ir.PX(qubits[i + 1]) for i in range(0, n - 1, 3)] +
                                        [ir.PY(qubits[1]) * ir.PZ(qubits[n - 1])])]
  results = {}
  for compat in (True, False):
    circ = models.DirectQuantumCircuit(hea_circuit(qubits, layers, "q"), tfq_compat_bit_order=compat)
    vals = np.random.default_rng(7).uniform(-1, 1, len(circ.symbol_names)).astype(np.float32)
    _set(circ.trainable_variables[0], vals)
    got = inference.AnalyticQuantumInference(circ).expectation(torch.from_numpy(bits), ops)
    weights = torch.as_tensor(np.arange(1, 7, dtype=np.float32).reshape(3, 2), device=got.device)
    (grad,) = torch.autograd.grad((got * weights).sum(), circ.trainable_variables)
    flat = circ.pqc.flat_gates(qubits, circ.symbol_names)
    masks = [op.masks(qubits) for op in ops]
    want, want_grad = C.expectation_vjp(n, flat, vals, O.apply_bit_order(bits, compat), masks, weights.cpu().numpy())
    np.testing.assert_allclose(got.detach().cpu().numpy(), want, atol=5e-5 * _norm(masks).max(), err_msg=f"compat={compat}")
    np.testing.assert_allclose(grad.cpu().numpy().reshape(-1), want_grad,
                               atol=3e-4 * max(1.0, float(np.abs(want_grad).max())), err_msg=f"compat={compat}")
    results[compat] = got.detach().cpu().numpy()
  assert np.abs(results[True] - results[False]).max() > 1e-2, "the two bit orders must differ on these inputs"


# ---- (d) the other entry points at 21..23 qubits (scripts/experiments/stress_api_sizes.py, one seed per size) ---------------
@pytest.mark.parametrize("n", [21, 22, 23])
def test_other_entry_points_at_21_to_23_qubits(n):
  rng = np.random.default_rng(9000 + n)
  gates, names = O.hea_gates(n, 3, "api")
  P = len(names)
  params = rng.uniform(-1, 1, P).astype(np.float32)
  ops = [O.tfim_ring_op(n), [(float(rng.normal()), 0, (1 << q) | (1 << ((q + 1) % n))) for q in range(n)] +
         [(float(rng.normal()), 0, 1 << q) for q in range(n)]]
  bits = rng.integers(0, 2, size=(2, n)).astype(np.int8)
  up = rng.normal(size=(2, len(ops))).astype(np.float32)
  eng = _engine(n, gates, P, ops)
  # final statevectors, global phase included (X / Z / CZ powers carry no global_shift)
  ref = C.statevector(n, gates, params, bits)
  sv = eng.statevector(bits, params).cpu().numpy().reshape(ref.shape)
  assert np.abs(sv - ref).max() < 5e-6
  # adjoint against the oracle on a masked parameter set, against parameter shift, and the retained pair
  want, want_grad = C.expectation_vjp(n, gates, params, bits, ops, up)
  mask = np.arange(P) < 6
  eng.set_gradient_mask(mask)
  v_a, g_a = eng.expectation_vjp(bits, params, up)
  v_s, g_s = eng.expectation_vjp(bits, params, up, method=E.GRAD_PARAMETER_SHIFT)
  eng.expectation(bits, params, retain=True)
  g_r = eng.expectation_vjp_retained(bits, params, up) if eng.retained is not None else g_a
  g_a, g_s, g_r = (g.cpu().numpy() for g in (g_a, g_s, g_r))
  scale = max(1.0, float(np.abs(want_grad).max()))
  assert (np.abs(v_a.cpu().numpy() - want) / _norm(ops)[None, :]).max() <= 5e-5
  np.testing.assert_allclose(g_a, np.where(mask, want_grad, 0.0), atol=3e-4 * scale, rtol=0)
  np.testing.assert_allclose(g_s, g_a, atol=5e-4 * scale, rtol=0)
  np.testing.assert_allclose(g_r, g_a, atol=1e-6 * scale, rtol=0)
  assert np.abs(v_a.cpu().numpy() - v_s.cpu().numpy()).max() < 5e-5 * n
  # Born-rule sampling: single-qubit marginals of 200k shots against |psi|^2
  shots = 200000
  s = eng.sample(bits[:1], params, shots, seed=n).cpu().numpy().reshape(shots, n)
  probs = np.abs(ref[0].astype(np.complex128)) ** 2
  idx = np.arange(1 << n)
  marg = np.array([probs[((idx >> (n - 1 - q)) & 1) == 1].sum() for q in range(n)])
  assert np.abs(s.mean(axis=0) - marg).max() < 6e-3
