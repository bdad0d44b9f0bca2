"""ctypes wrapper of oracle/qhbm_cpu.c (TEST INFRASTRUCTURE / CPU baseline only)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (QHBM_ORACLE_LIB: a test hook -- tests/test_bench_gpu.py hides the library to see bench.py fail loudly)
LIB_PATH = os.environ.get("QHBM_ORACLE_LIB") or os.path.join(_HERE, "libqhbm_cpu.so")


class _Gate(ctypes.Structure):
  _fields_ = [("kind", ctypes.c_int32), ("q0", ctypes.c_int32), ("q1", ctypes.c_int32),
              ("param_idx", ctypes.c_int32), ("scalar", ctypes.c_float), ("offset", ctypes.c_float)]


_lib = None


def _load():
  global _lib
  if _lib is None:
    _lib = ctypes.CDLL(LIB_PATH)
    _lib.qo_max_threads.restype = ctypes.c_int
  return _lib


def max_threads():
  return _load().qo_max_threads()


def _pack(n, gates, params, bits, ops):
  arr = (_Gate * max(len(gates), 1))()
  for i, g in enumerate(gates):   # a 7th entry (cirq global_shift) never changes an expectation value
    kind, q0, q1, pidx, scalar, offset = g[:6]
    arr[i] = _Gate(int(kind), int(q0), int(q1), int(pidx), float(scalar), float(offset))
  offsets, coeffs, xs, zs = [0], [], [], []
  for op in ops:
    for c, x, z in op:
      coeffs.append(c); xs.append(x); zs.append(z)
    offsets.append(len(coeffs))
  return (arr, np.ascontiguousarray(params, dtype=np.float32),
          np.ascontiguousarray(bits, dtype=np.int8), np.asarray(offsets, np.int32),
          np.asarray(coeffs, np.float32), np.asarray(xs, np.uint64), np.asarray(zs, np.uint64))


def expectation(n, gates, params, bits, ops, n_threads=0):
  lib = _load()
  arr, p, b, off, cf, xm, zm = _pack(n, gates, params, bits, ops)
  out = np.zeros((b.shape[0], len(ops)), np.float32)
  vp = ctypes.c_void_p
  lib.qo_expectation(ctypes.c_int(n), ctypes.c_int(len(gates)), arr, vp(p.ctypes.data), vp(b.ctypes.data),
                     ctypes.c_int(b.shape[0]), ctypes.c_int(len(ops)), vp(off.ctypes.data),
                     vp(cf.ctypes.data), vp(xm.ctypes.data), vp(zm.ctypes.data), vp(out.ctypes.data),
                     ctypes.c_int(n_threads))
  return out


def expectation_vjp(n, gates, params, bits, ops, upstream, n_threads=0):
  lib = _load()
  arr, p, b, off, cf, xm, zm = _pack(n, gates, params, bits, ops)
  up = np.ascontiguousarray(upstream, dtype=np.float32)
  vals = np.zeros((b.shape[0], len(ops)), np.float32)
  grad = np.zeros((len(p),), np.float32)
  vp = ctypes.c_void_p
  lib.qo_expectation_vjp(ctypes.c_int(n), ctypes.c_int(len(gates)), arr, vp(p.ctypes.data),
                         vp(b.ctypes.data), ctypes.c_int(b.shape[0]), ctypes.c_int(len(ops)),
                         vp(off.ctypes.data), vp(cf.ctypes.data), vp(xm.ctypes.data), vp(zm.ctypes.data),
                         vp(up.ctypes.data), vp(vals.ctypes.data), vp(grad.ctypes.data),
                         ctypes.c_int(len(p)), ctypes.c_int(n_threads))
  return vals, grad


def statevector(n, gates, params, bits, n_threads=0):
  """complex64 [batch, 2^n] final states (qubit 0 = most significant index bit); cirq's global_shift is not carried:
  exact for X / Z / CZ powers."""
  lib = _load()
  arr, p, b, _, _, _, _ = _pack(n, gates, params, bits, [])
  out = np.zeros((b.shape[0], 1 << n), np.complex64)
  vp = ctypes.c_void_p
  lib.qo_statevector(ctypes.c_int(n), ctypes.c_int(len(gates)), arr, vp(p.ctypes.data), vp(b.ctypes.data),
                     ctypes.c_int(b.shape[0]), vp(out.ctypes.data), ctypes.c_int(n_threads))
  return out


def expectation_vjp_diag(n, gates, params, bits, ops, upstream=None, n_threads=0):
  """The timed baseline path (oracle/qhbm_cpu_diag.c: merged diagonal runs, AVX2 one-qubit kernel, fused adjoint steps):
  (values, grad), or (values, None) when `upstream` is None (forward only).  Same contract as `expectation_vjp`."""
  lib = _load()
  arr, p, b, off, cf, xm, zm = _pack(n, gates, params, bits, ops)
  vals = np.zeros((b.shape[0], len(ops)), np.float32)
  vp = ctypes.c_void_p
  up = grad = None
  if upstream is not None:
    up = np.ascontiguousarray(upstream, dtype=np.float32)
    grad = np.zeros((len(p),), np.float32)
  lib.qo_expectation_vjp_diag(ctypes.c_int(n), ctypes.c_int(len(gates)), arr, vp(p.ctypes.data), vp(b.ctypes.data),
                              ctypes.c_int(b.shape[0]), ctypes.c_int(len(ops)), vp(off.ctypes.data), vp(cf.ctypes.data),
                              vp(xm.ctypes.data), vp(zm.ctypes.data), vp(up.ctypes.data) if up is not None else None,
                              vp(vals.ctypes.data), vp(grad.ctypes.data) if grad is not None else None,
                              ctypes.c_int(len(p)), ctypes.c_int(n_threads))
  return vals, grad
