"""The C ABI without torch: `examples/ctypes_binding.py` (the helper code INTEGRATION.md gives a
reference maintainer) drives the engine with ctypes + numpy + hipMalloc only."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import qhbm_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _module():
  spec = importlib.util.spec_from_file_location("ctypes_binding", os.path.join(ROOT, "examples", "ctypes_binding.py"))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def test_closed_form_self_check():
  _module().main()


def test_values_and_vjp_against_oracle_through_plain_ctypes():
  cb = _module()
  hip, lib = cb.load()
  rng = np.random.default_rng(8)
  n, layers = 11, 2
  gates, names = O.hea_gates(n, layers, "c")
  params = rng.uniform(-1, 1, len(names))
  ops = [O.xxz_chain_op(n), O.tfim_ring_op(n)]
  bits = rng.integers(0, 2, size=(5, n)).astype(np.int8)
  up = rng.normal(size=(5, 2)).astype(np.float32)
  handle = cb.make_engine(lib, n, gates, len(names), ops)
  vals, grad = cb.expectation_and_vjp(hip, lib, handle, bits, params, up)
  want, jac = O.expectation_jacobian(n, gates, params, bits, ops)
  np.testing.assert_allclose(vals, want, atol=1e-4)
  want_grad = np.einsum("bt,btp->p", up, jac)
  np.testing.assert_allclose(grad, want_grad, atol=1e-4 * max(1.0, np.abs(want_grad).max()))
  lib.qhbm_destroy(handle)
