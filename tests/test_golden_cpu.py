"""The oracles against the committed golden vectors (CPU)."""
import os

import numpy as np
import pytest

from oracle import qhbm_cpu as C
from oracle import qhbm_oracle as O
from tests import golden_util as G


@pytest.mark.parametrize("name", G.hea_files() + ["all_kinds_n5.npz"])
def test_numpy_oracle_reproduces_golden(name):
  g = G.load(name)
  n, gates, ops = int(g["n"]), G.gates_of(g["gates"]), G.ops_of(g["ops"])
  vals, jac = O.expectation_jacobian(n, gates, g["params"], g["bits"], ops)
  np.testing.assert_allclose(vals, g["values"], atol=1e-12)
  np.testing.assert_allclose(jac, g["jacobian"], atol=1e-11)


@pytest.mark.skipif(not os.path.exists(C.LIB_PATH), reason="oracle/libqhbm_cpu.so not built")
@pytest.mark.parametrize("name", G.hea_files() + ["all_kinds_n5.npz"])
def test_c_oracle_reproduces_golden(name):
  g = G.load(name)
  n, gates, ops = int(g["n"]), G.gates_of(g["gates"]), G.ops_of(g["ops"])
  np.testing.assert_allclose(C.expectation(n, gates, g["params"], g["bits"], ops), g["values"], atol=3e-5)
  up = np.ones_like(g["values"])
  _, grad = C.expectation_vjp(n, gates, g["params"], g["bits"], ops, up)
  want = np.einsum("bt,btp->p", up, g["jacobian"])
  np.testing.assert_allclose(grad, want, atol=3e-4 * max(1.0, np.abs(want).max()))


def test_bit_order_fixture():
  g = G.load("hea_n12_bit_order.npz")
  n, gates, ops = int(g["n"]), G.gates_of(g["gates"]), G.ops_of(g["ops"])
  assert g["tfq_permutation"].tolist() == [0, 1, 10, 11, 2, 3, 4, 5, 6, 7, 8, 9]
  np.testing.assert_allclose(O.expectation(n, gates, g["params"], g["bits"], ops, False), g["values_direct"], atol=1e-12)
  np.testing.assert_allclose(O.expectation(n, gates, g["params"], g["bits"], ops, True), g["values_tfq_compat"], atol=1e-12)
  assert np.abs(g["values_direct"] - g["values_tfq_compat"]).max() > 1e-3


def test_vqt_fixture():
  g = G.load("vqt_c1.npz")
  n, gates = int(g["n"]), G.gates_of(g["gates"])
  thetas = g["thetas"]
  loss, dtheta, dparams = O.vqt_loss_and_grads(
      n, gates, g["params"], g["samples"], G.ops_of(g["target"])[0], float(g["beta"]),
      lambda b: O.bernoulli_energy(b, thetas), O.spins_from_bitstrings, float(g["log_partition"]))
  np.testing.assert_allclose(loss, float(g["loss"]), atol=1e-12)
  np.testing.assert_allclose(dtheta, g["dtheta"], atol=1e-12)
  np.testing.assert_allclose(dparams, g["dparams"], atol=1e-12)
