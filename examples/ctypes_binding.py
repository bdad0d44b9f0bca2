"""The C ABI driven with nothing but ctypes + numpy (no torch): the helpers a reference-side
binding needs (INTEGRATION.md section 2 uses exactly these) and a small self-check.

    python examples/ctypes_binding.py          # on an MI355X box, after __graft_entry__.build()

Device memory comes straight from the HIP runtime (`libamdhip64.so`: hipMalloc / hipMemcpy /
hipFree); every entry point of `include/qhbm_engine.h` takes plain pointers and sizes.
"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENGINE_LIB = os.environ.get("QHBM_ENGINE_LIB") or os.path.join(ROOT, "qhbm-library_amd", "lib", "libqhbm_engine.so")

_H2D, _D2H = 1, 2  # hipMemcpyHostToDevice, hipMemcpyDeviceToHost


class Gate(ctypes.Structure):  # struct qhbm_gate
  _fields_ = [("kind", ctypes.c_int32), ("q0", ctypes.c_int32), ("q1", ctypes.c_int32),
              ("param_idx", ctypes.c_int32), ("scalar", ctypes.c_float), ("offset", ctypes.c_float),
              ("global_shift", ctypes.c_float)]  # ABI v3: cirq's EigenGate global_shift (0 when omitted)


def load():
  hip = ctypes.CDLL("libamdhip64.so")
  hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
  hip.hipFree.argtypes = [ctypes.c_void_p]
  hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
  lib = ctypes.CDLL(ENGINE_LIB)
  lib.qhbm_last_error.argtypes = [ctypes.c_void_p]
  lib.qhbm_last_error.restype = ctypes.c_char_p
  vp, i32 = ctypes.c_void_p, ctypes.c_int
  lib.qhbm_create.argtypes = [i32, ctypes.POINTER(vp)]
  lib.qhbm_destroy.argtypes = [vp]
  lib.qhbm_destroy.restype = None
  lib.qhbm_set_circuit.argtypes = [vp, i32, i32, ctypes.POINTER(Gate), i32]
  lib.qhbm_set_observables.argtypes = [vp, i32, vp, vp, vp, vp]
  lib.qhbm_expectation.argtypes = [vp, vp, i32, vp, vp, vp]
  lib.qhbm_expectation_retain.argtypes = [vp, vp, i32, vp, vp, vp]
  lib.qhbm_expectation_vjp.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32, vp]
  lib.qhbm_expectation_vjp_retained.argtypes = [vp, vp, i32, vp, vp, vp, vp]
  return hip, lib


class DeviceBuffer:
  """A hipMalloc'ed array with numpy upload / download."""

  def __init__(self, hip, shape, dtype):
    self.hip, self.shape, self.dtype = hip, tuple(shape), np.dtype(dtype)
    self.nbytes = max(1, int(np.prod(self.shape)) * self.dtype.itemsize)
    self.ptr = ctypes.c_void_p()
    if hip.hipMalloc(ctypes.byref(self.ptr), self.nbytes) != 0:
      raise MemoryError(f"hipMalloc({self.nbytes})")

  @classmethod
  def from_host(cls, hip, array, dtype):
    host = np.ascontiguousarray(array, dtype=dtype)
    buf = cls(hip, host.shape, dtype)
    if host.size and hip.hipMemcpy(buf.ptr, host.ctypes.data, host.nbytes, _H2D) != 0:
      raise RuntimeError("hipMemcpy H2D")
    return buf

  def to_host(self):
    out = np.empty(self.shape, self.dtype)
    if out.size and self.hip.hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, _D2H) != 0:  # synchronises the null stream
      raise RuntimeError("hipMemcpy D2H")
    return out

  def __del__(self):
    if getattr(self, "ptr", None):
      self.hip.hipFree(self.ptr)
      self.ptr = None


def check(lib, handle, rc):
  if rc != 0:
    raise RuntimeError(lib.qhbm_last_error(handle).decode())


def make_engine(lib, n_qubits, gates, n_params, ops, device=0):
  """gates: [(kind, q0, q1, param_idx, scalar, offset[, global_shift])]; ops: [[(coeff, x_mask, z_mask)]]."""
  handle = ctypes.c_void_p()
  check(lib, None, lib.qhbm_create(device, ctypes.byref(handle)))
  arr = (Gate * max(1, len(gates)))(*[Gate(*g) for g in gates])
  check(lib, handle, lib.qhbm_set_circuit(handle, n_qubits, len(gates), arr, n_params))
  offsets = np.cumsum([0] + [len(op) for op in ops]).astype(np.int32)
  flat = [t for op in ops for t in op]
  coeffs = np.array([t[0] for t in flat], np.float32)
  xs = np.array([t[1] for t in flat], np.uint64)
  zs = np.array([t[2] for t in flat], np.uint64)
  check(lib, handle, lib.qhbm_set_observables(handle, len(ops), offsets.ctypes.data, coeffs.ctypes.data,
                                               xs.ctypes.data, zs.ctypes.data))
  return handle


def expectation_and_vjp(hip, lib, handle, bits, params, upstream):
  """values [U, T] and grad [P] = sum upstream * d values / d params: forward now, backward from
  the retained states (what a tf.custom_gradient / autograd wrapper does in two steps)."""
  bits = np.ascontiguousarray(bits, np.int8)
  n_states, n_ops = bits.shape[0], np.shape(upstream)[1]
  d_bits = DeviceBuffer.from_host(hip, bits, np.int8)
  d_params = DeviceBuffer.from_host(hip, params, np.float32)
  d_out = DeviceBuffer(hip, (n_states, n_ops), np.float32)
  check(lib, handle, lib.qhbm_expectation_retain(handle, d_bits.ptr, n_states, d_params.ptr, d_out.ptr, None))
  d_up = DeviceBuffer.from_host(hip, upstream, np.float32)
  d_grad = DeviceBuffer(hip, (len(params),), np.float32)
  if lib.qhbm_expectation_vjp_retained(handle, d_bits.ptr, n_states, d_params.ptr, d_up.ptr, d_grad.ptr, None) != 0:
    # nothing retained (batch larger than one backward chunk): simulate again
    check(lib, handle, lib.qhbm_expectation_vjp(handle, d_bits.ptr, n_states, d_params.ptr, d_up.ptr, None,
                                                d_grad.ptr, 0, None))
  return d_out.to_host(), d_grad.to_host()


def main():
  hip, lib = load()
  # two qubits: X(q)**p on both (tests/inference/qnn_test.py:83-180), measure Z0 and Y1
  gates = [(1, 0, -1, 0, 1.0, 0.0), (1, 1, -1, 0, 1.0, 0.0)]
  ops = [[(1.0, 0, 1)], [(1.0, 2, 2)]]
  handle = make_engine(lib, 2, gates, 1, ops)
  bits = np.array([[0, 0], [1, 0], [0, 1], [1, 1]], np.int8)
  p = 0.37
  vals, grad = expectation_and_vjp(hip, lib, handle, bits, [p], np.ones((4, 2), np.float32))
  want = np.array([[(-1.0)**b0 * np.cos(np.pi * p), -(-1.0)**b1 * np.sin(np.pi * p)] for b0, b1 in bits])
  print("values\n", vals, "\nclosed form\n", want, "\ngrad", grad)
  assert np.abs(vals - want).max() < 1e-5
  lib.qhbm_destroy(handle)


if __name__ == "__main__":
  main()
