"""Values and gradients under the planner's OWN choices at the qubit counts between the fixtures (15..19): tile sizes, a
wide last pass, measurement-only passes and the measuring kernel change with the size, and a wrong-value bug once lived at
19 qubits only (HISTORY round 5).  One seed per size here; scripts/experiments/stress_{measure,default_plans,api_sizes}.py
are the long versions."""
import numpy as np
import pytest

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


@pytest.mark.parametrize("n", [15, 16, 17, 18, 19])
def test_measured_values_of_many_diagonal_terms_under_default_plans(n):
  """Short-range Z strings (what the last pass of the sweep measures itself) + random ones (measurement-only passes), below
  and above the Walsh-Hadamard threshold, as shards / a few observables / one sum; three plan option sets."""
  rng = np.random.default_rng(1500 + n)
  gates, names = O.hea_gates(n, 3, "dp")
  params = rng.uniform(-1, 1, len(names))
  chain = [(float(rng.normal()), 0, (1 << q) | (1 << ((q + 1) % n))) for q in range(n)]
  chain += [(float(rng.normal()), 0, 1 << q) for q in range(n)]
  scattered = [(float(rng.normal()), 0, int(sum(1 << int(q) for q in rng.choice(n, size=int(rng.integers(1, 4)), replace=False))))
               for _ in range(20)]
  flips = [(float(rng.normal()), 1 << int(q), 0) for q in rng.choice(n, size=3, replace=False)]
  layouts = {"shards": [[t] for t in chain + scattered] + [flips],
             "few": [chain[0::3] + flips[:1], chain[1::3] + scattered[:7], chain[2::3] + scattered[7:]],
             "two": [chain + scattered + flips, chain[: n // 2]]}
  bits = rng.integers(0, 2, size=(2, n)).astype(np.int8)
  for name, ops in layouts.items():
    want = O.expectation(n, gates, params, bits, ops)
    norm = np.maximum(np.array([sum(abs(c) for c, _, _ in op) for op in ops]), 1.0)
    for opts in ({}, {"observable_kernel": 0, "multi_observable_values": 0}, {"wide_last_pass": 0}):
      got = _engine(n, gates, len(names), ops, **opts).expectation(bits, params).cpu().numpy()
      err = np.abs(got - want) / norm[None, :]
      assert err.max() <= 5e-5, (name, opts, float(err.max()))


@pytest.mark.parametrize("n", [15, 16, 17])
def test_values_and_gradients_under_default_plans_and_option_sets(n):
  rng = np.random.default_rng(1700 + n)
  gates, names = O.hea_gates(n, 2 + n % 2, "dg")
  P = len(names)
  params = rng.uniform(-1, 1, P)
  ops = [O.random_pauli_op(n, 12, n, p_identity=0.75), O.xxz_chain_op(n)][: 1 + n % 2]
  bits = rng.integers(0, 2, size=(2, n)).astype(np.int8)
  up = rng.normal(size=(2, len(ops)))
  want_vals, want_jac = O.expectation_jacobian(n, gates, params, bits, ops)
  want_grad = np.einsum("bt,btp->p", up, want_jac)
  norm = np.maximum(np.array([sum(abs(c) for c, _, _ in op) for op in ops]), 1.0)
  mask = rng.random(P) < 0.7
  for opts in ({}, {"adjoint_relabel": 0}, {"forward_pairs": 0, "wide_last_pass": 0}, {"adjoint_tile_qubits": 13}):
    for use_mask in (False, True):
      eng = _engine(n, gates, P, ops, **opts)
      if use_mask:
        eng.set_gradient_mask(mask)
      vals, grad = eng.expectation_vjp(bits, params, up)
      assert (np.abs(vals.cpu().numpy() - want_vals) / norm[None, :]).max() <= 5e-5, (opts, use_mask)
      wg = np.where(mask, want_grad, 0.0) if use_mask else want_grad
      np.testing.assert_allclose(grad.cpu().numpy(), wg, atol=3e-4 * max(1.0, float(np.abs(wg).max())), rtol=0,
                                 err_msg=f"{opts} mask={use_mask}")
