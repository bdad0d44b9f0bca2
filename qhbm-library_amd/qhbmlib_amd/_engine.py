"""ctypes binding of the HIP engine's C ABI (include/qhbm_engine.h).

There is no CPU fallback: if `libqhbm_engine.so` is missing or no GPU is
present, every compute call raises.  torch is used for device memory and
streams only.
"""
import ctypes
import os
import warnings

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QHBM_ENGINE_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libqhbm_engine.so")

# Gate kinds: values of enum qhbm_gate_kind.
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

ABI_VERSION = 5  # include/qhbm_engine.h QHBM_ABI_VERSION

GRAD_ADJOINT = 0
GRAD_PARAMETER_SHIFT = 1

ABI_SYMBOLS = (
    "qhbm_abi_version", "qhbm_create", "qhbm_destroy", "qhbm_last_error",
    "qhbm_set_circuit", "qhbm_set_gradient_mask", "qhbm_set_observables", "qhbm_set_option",
    "qhbm_workspace_bytes", "qhbm_allocated_bytes", "qhbm_expectation", "qhbm_expectation_vjp",
    "qhbm_expectation_retain", "qhbm_expectation_vjp_retained", "qhbm_retained_states", "qhbm_state_gradients",
    "qhbm_expectation_jacobian", "qhbm_statevector", "qhbm_sample", "qhbm_sample_counts", "qhbm_parity_energy", "qhbm_parity_energy_vjp",
    "qhbm_num_passes", "qhbm_describe_schedule",
    "qhbm_kernel_time_ms", "qhbm_traffic_model", "qhbm_flop_model", "qhbm_op_census", "qhbm_clock_probe", "qhbm_plan_builds",
)


class QhbmGate(ctypes.Structure):
  _fields_ = [("kind", ctypes.c_int32), ("q0", ctypes.c_int32),
              ("q1", ctypes.c_int32), ("param_idx", ctypes.c_int32),
              ("scalar", ctypes.c_float), ("offset", ctypes.c_float),
              ("global_shift", ctypes.c_float)]


class EngineError(RuntimeError):
  pass


_lib = None


def load_library():
  """Loads libqhbm_engine.so (built by `__graft_entry__.build()`)."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise EngineError(
        f"{LIB_PATH} not found: build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()'); "
        "there is no CPU fallback.")
  lib = ctypes.CDLL(LIB_PATH)
  vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
  lib.qhbm_abi_version.restype = i32
  found = lib.qhbm_abi_version()
  if found != ABI_VERSION:
    # an OLDER library handed in on purpose (QHBM_ENGINE_LIB, the one-box A/B runs of scripts/r05_ab.sh) is tolerated
    # with a warning: the optional symbols below are skipped; anything else is a stale build
    if os.environ.get("QHBM_ENGINE_LIB") and found < ABI_VERSION:
      warnings.warn(f"{LIB_PATH} exports ABI v{found}, this binding is written for v{ABI_VERSION}: newer entry "
                    "points are unavailable")
    else:
      raise EngineError(f"{LIB_PATH} exports ABI v{found}, this binding needs v{ABI_VERSION}: rebuild the engine "
                        "(python -c 'import __graft_entry__ as g; g.build()')")
  lib.qhbm_create.argtypes = [i32, ctypes.POINTER(vp)]
  lib.qhbm_destroy.argtypes = [vp]
  lib.qhbm_destroy.restype = None
  lib.qhbm_last_error.argtypes = [vp]
  lib.qhbm_last_error.restype = ctypes.c_char_p
  lib.qhbm_set_circuit.argtypes = [vp, i32, i32, ctypes.POINTER(QhbmGate), i32]
  lib.qhbm_set_gradient_mask.argtypes = [vp, vp, i32]
  lib.qhbm_set_observables.argtypes = [vp, i32, vp, vp, vp, vp]
  lib.qhbm_set_option.argtypes = [vp, ctypes.c_char_p, i64]
  lib.qhbm_workspace_bytes.argtypes = [vp, i32, i32,
                                       ctypes.POINTER(ctypes.c_size_t)]
  lib.qhbm_allocated_bytes.argtypes = [vp, ctypes.POINTER(ctypes.c_size_t)]
  lib.qhbm_retained_states.argtypes = [vp, ctypes.POINTER(i32)]
  lib.qhbm_expectation.argtypes = [vp, vp, i32, vp, vp, vp]
  lib.qhbm_expectation_vjp.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32, vp]
  lib.qhbm_expectation_retain.argtypes = [vp, vp, i32, vp, vp, vp]
  lib.qhbm_expectation_vjp_retained.argtypes = [vp, vp, i32, vp, vp, vp, vp]
  lib.qhbm_state_gradients.argtypes = [vp, i32, vp, vp]
  lib.qhbm_expectation_jacobian.argtypes = [vp, vp, i32, vp, vp, vp, vp]
  lib.qhbm_statevector.argtypes = [vp, vp, i32, vp, vp, vp]
  lib.qhbm_sample.argtypes = [vp, vp, i32, vp, i32, ctypes.c_uint64, i32, ctypes.c_double, vp, vp]
  lib.qhbm_sample_counts.argtypes = [vp, vp, i32, vp, i32, vp, vp, i32, ctypes.c_uint64, vp, vp]
  lib.qhbm_parity_energy.argtypes = [vp, i64, i32, vp, vp, i32, vp, vp]
  lib.qhbm_parity_energy_vjp.argtypes = [vp, i64, i32, vp, i32, vp, vp, vp]
  lib.qhbm_num_passes.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32)]
  lib.qhbm_describe_schedule.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t]
  lib.qhbm_kernel_time_ms.argtypes = [
      vp, i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64),
      ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64),
      ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64)]
  lib.qhbm_traffic_model.argtypes = [vp, i32, i32] + [ctypes.POINTER(ctypes.c_double)] * 3
  lib.qhbm_flop_model.argtypes = [vp, i32, i32] + [ctypes.POINTER(ctypes.c_double)] * 3
  try:
    lib.qhbm_op_census.argtypes = [vp, i32, i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i32)]
    lib.qhbm_plan_builds.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    lib.qhbm_clock_probe.argtypes = [vp] + [ctypes.POINTER(ctypes.c_double)] * 3 + [vp]
  except AttributeError:  # an older library given through QHBM_ENGINE_LIB (A/B runs): the probe is optional there
    pass
  _lib = lib
  return lib


def _check_global(rc):
  if rc != 0:
    raise EngineError(load_library().qhbm_last_error(None).decode())


class _ParityEnergyFunction(torch.autograd.Function):
  """energies[i] = sum_k thetas[k] * parity_k(bits[i]) on the GPU (qhbm_parity_energy); the
  backward is qhbm_parity_energy_vjp.  `masks` is an int64 CUDA tensor of column masks."""

  @staticmethod
  def forward(ctx, thetas, bits, masks):
    lib = load_library()
    bits = bits.to(torch.int8).contiguous()
    th = thetas.detach().to(device=bits.device, dtype=torch.float32).contiguous()
    out = torch.empty((bits.shape[0],), dtype=torch.float32, device=bits.device)
    with torch.cuda.device(bits.device):
      _check_global(lib.qhbm_parity_energy(
          bits.data_ptr(), bits.shape[0], bits.shape[1], masks.data_ptr(), th.data_ptr(),
          masks.numel(), out.data_ptr(),
          ctypes.c_void_p(torch.cuda.current_stream(bits.device).cuda_stream)))
    ctx.bits, ctx.masks = bits, masks
    ctx.theta_device = thetas.device
    return out

  @staticmethod
  def backward(ctx, upstream):
    lib = load_library()
    bits, masks = ctx.bits, ctx.masks
    w = upstream.to(device=bits.device, dtype=torch.float32).contiguous()
    grad = torch.empty((masks.numel(),), dtype=torch.float32, device=bits.device)
    with torch.cuda.device(bits.device):
      _check_global(lib.qhbm_parity_energy_vjp(
          bits.data_ptr(), bits.shape[0], bits.shape[1], masks.data_ptr(), masks.numel(),
          w.data_ptr(), grad.data_ptr(),
          ctypes.c_void_p(torch.cuda.current_stream(bits.device).cuda_stream)))
    return grad.to(ctx.theta_device), None, None


def parity_energy(thetas, bits, masks):
  """Differentiable (w.r.t. `thetas`) spin-parity energies of CUDA `bits` [N, n]."""
  if not bits.is_cuda:
    raise EngineError("parity_energy runs on the GPU: pass CUDA bitstrings (CPU tensors use the torch layers)")
  return _ParityEnergyFunction.apply(thetas, bits, masks)


class Engine:
  """One engine per device.  `device=None` makes a planning-only engine
  (schedules can be inspected, compute calls raise)."""

  def __init__(self, device=0):
    self._lib = load_library()
    self._h = ctypes.c_void_p()
    dev = -1 if device is None else int(device)
    if self._lib.qhbm_create(dev, ctypes.byref(self._h)) != 0:
      raise EngineError(self._lib.qhbm_last_error(None).decode())
    self.device = None if device is None else torch.device("cuda", dev)
    self.n_qubits = 0
    self.n_params = 0
    self.n_ops = 0
    self.retained = None

  def close(self):
    if getattr(self, "_h", None) is not None and self._h:
      self._lib.qhbm_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  def _check(self, rc):
    if rc != 0:
      raise EngineError(self._lib.qhbm_last_error(self._h).decode())

  # ---- model -------------------------------------------------------------
  def set_circuit(self, n_qubits, gates, n_params):
    """gates: iterable of (kind, q0, q1, param_idx, scalar, offset[, global_shift])."""
    gates = list(gates)
    arr = (QhbmGate * max(len(gates), 1))()
    for i, g in enumerate(gates):
      kind, q0, q1, pidx, scalar, offset = g[:6]
      arr[i] = QhbmGate(int(kind), int(q0), int(q1), int(pidx), float(scalar),
                        float(offset), float(g[6]) if len(g) > 6 else 0.0)
    self._check(
        self._lib.qhbm_set_circuit(self._h, int(n_qubits), len(gates), arr,
                                   int(n_params)))
    if n_qubits != self.n_qubits:
      self.n_ops = 0
    self.n_qubits, self.n_params = int(n_qubits), int(n_params)
    self._grad_mask = None  # (qhbm_set_circuit resets the engine's mask)

  def set_observables(self, ops):
    """ops: list of ops, each a list of (coeff, x_mask, z_mask), qubit space."""
    offsets = [0]
    coeffs, xs, zs = [], [], []
    for op in ops:
      for coeff, x, z in op:
        coeffs.append(coeff)
        xs.append(x)
        zs.append(z)
      offsets.append(len(coeffs))
    off = np.asarray(offsets, dtype=np.int32)
    cf = np.asarray(coeffs, dtype=np.float32)
    xm = np.asarray(xs, dtype=np.uint64)
    zm = np.asarray(zs, dtype=np.uint64)
    self._check(
        self._lib.qhbm_set_observables(self._h, len(ops), off.ctypes.data,
                                       cf.ctypes.data, xm.ctypes.data,
                                       zm.ctypes.data))
    self.n_ops = len(ops)

  def set_gradient_mask(self, needs_grad):
    """needs_grad: one truth value per parameter (None: all).  Frozen parameters get zero gradient entries and no
    gradient work; the adjoint sweep stops at the first gate of a parameter that is not frozen
    (include/qhbm_engine.h qhbm_set_gradient_mask).  Reset by set_circuit."""
    key = None if needs_grad is None else tuple(bool(f) for f in needs_grad)
    if key == getattr(self, "_grad_mask", None):
      return
    if key is None:
      self._check(self._lib.qhbm_set_gradient_mask(self._h, None, 0))
    else:
      mask = np.ascontiguousarray(np.asarray(key, dtype=np.uint8))
      if mask.shape != (self.n_params,):
        raise EngineError(f"gradient mask of shape {mask.shape} for {self.n_params} parameters")
      self._check(self._lib.qhbm_set_gradient_mask(self._h, mask.ctypes.data, self.n_params))
    self._grad_mask = key

  def set_option(self, name, value):
    self._check(self._lib.qhbm_set_option(self._h, name.encode(), int(value)))

  # ---- introspection -------------------------------------------------------
  def num_passes(self):
    f, b = ctypes.c_int(), ctypes.c_int()
    self._check(self._lib.qhbm_num_passes(self._h, ctypes.byref(f),
                                          ctypes.byref(b)))
    return f.value, b.value

  def describe_schedule(self):
    buf = ctypes.create_string_buffer(1 << 16)
    self._check(self._lib.qhbm_describe_schedule(self._h, buf, len(buf)))
    return buf.value.decode()

  def workspace_bytes(self, num_states, with_vjp=False):
    out = ctypes.c_size_t()
    self._check(
        self._lib.qhbm_workspace_bytes(self._h, int(num_states), int(with_vjp),
                                       ctypes.byref(out)))
    return out.value

  def allocated_bytes(self):
    """Device memory this engine holds right now (workspace + gradient partials)."""
    out = ctypes.c_size_t()
    self._check(self._lib.qhbm_allocated_bytes(self._h, ctypes.byref(out)))
    return out.value

  def retained_states(self):
    """Number of final states the workspace keeps for `expectation_vjp_retained` (0: none)."""
    out = ctypes.c_int()
    self._check(self._lib.qhbm_retained_states(self._h, ctypes.byref(out)))
    return out.value

  def kernel_time_ms(self, reset=True):
    f, b, o = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    nf, nb, no = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    self._check(
        self._lib.qhbm_kernel_time_ms(self._h, int(reset), ctypes.byref(f),
                                      ctypes.byref(nf), ctypes.byref(b),
                                      ctypes.byref(nb), ctypes.byref(o), ctypes.byref(no)))
    return {"fwd_ms": f.value, "fwd_launches": nf.value, "bwd_ms": b.value,
            "bwd_launches": nb.value, "obs_ms": o.value, "obs_launches": no.value}

  def traffic_model(self, num_states, with_vjp=True):
    """HBM bytes one call must move (every touched tile read and written once): dict of
    forward / lambda = O psi / adjoint bytes (include/qhbm_engine.h qhbm_traffic_model)."""
    f, o, b = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    self._check(self._lib.qhbm_traffic_model(self._h, int(num_states), int(with_vjp), ctypes.byref(f),
                                             ctypes.byref(o), ctypes.byref(b)))
    return {"fwd_bytes": f.value, "obs_bytes": o.value, "bwd_bytes": b.value}

  def flop_model(self, num_states, with_vjp=True):
    """fp32 operations (FMA = 2) of the gate arithmetic of one call: dict of forward / lambda = O psi /
    adjoint flops (include/qhbm_engine.h qhbm_flop_model)."""
    f, o, b = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    self._check(self._lib.qhbm_flop_model(self._h, int(num_states), int(with_vjp), ctypes.byref(f),
                                          ctypes.byref(o), ctypes.byref(b)))
    return {"fwd_flops": f.value, "obs_flops": o.value, "bwd_flops": b.value}

  def plan_builds(self):
    """(forward, backward) plan searches run so far (include/qhbm_engine.h qhbm_plan_builds)."""
    f, b = ctypes.c_int64(), ctypes.c_int64()
    self._check(self._lib.qhbm_plan_builds(self._h, ctypes.byref(f), ctypes.byref(b)))
    return f.value, b.value

  def clock_probe(self):
    """The chip's sustained packed-fp32 rate right now: dict of `ghz` (shader clock during the probe),
    `cycles_per_pk_fma` and `tflops` (include/qhbm_engine.h qhbm_clock_probe).  Synchronises the stream."""
    if self.device is None:
      raise EngineError("planning-only engine: no device, no CPU fallback")
    g, c, t = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    self._check(self._lib.qhbm_clock_probe(self._h, ctypes.byref(g), ctypes.byref(c), ctypes.byref(t), self._stream()))
    return {"ghz": g.value, "cycles_per_pk_fma": c.value, "tflops": t.value}

  CENSUS_COLUMNS = ("tiles", "rounds", "rounds_barrier", "rounds_no_barrier", "instances", "x", "x_no_slot", "full",
                    "ph1", "ph2", "cph_tile_on", "cph_wave_on", "cph_lane", "cph_off", "reduce8")

  def op_census(self, adjoint=True, max_passes=64):
    """Executed micro-ops per pass in wave-executions per state (include/qhbm_engine.h qhbm_op_census):
    a list of dicts, one per pass of the forward or backward schedule."""
    import numpy as np  # pylint: disable=import-outside-toplevel
    ncol = len(self.CENSUS_COLUMNS)
    out = np.zeros((max_passes, ncol), np.float64)
    n = ctypes.c_int()
    self._check(self._lib.qhbm_op_census(self._h, int(bool(adjoint)), max_passes,
                                         out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.byref(n)))
    return [dict(zip(self.CENSUS_COLUMNS, out[i])) for i in range(min(n.value, max_passes))]

  # ---- hot path --------------------------------------------------------------
  def _prep(self, bits, params):
    if self.device is None:
      raise EngineError("planning-only engine: no device, no CPU fallback")
    bits = torch.as_tensor(bits).to(device=self.device, dtype=torch.int8)
    bits = bits.contiguous()
    if bits.dim() != 2 or bits.shape[1] != self.n_qubits:
      raise ValueError(
          f"bitstrings must have shape [batch, {self.n_qubits}], got "
          f"{tuple(bits.shape)}")
    params = torch.as_tensor(params).to(device=self.device,
                                        dtype=torch.float32).contiguous()
    if params.numel() != self.n_params:
      raise ValueError(f"expected {self.n_params} parameters")
    return bits, params

  def _stream(self):
    return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

  def expectation(self, bits, params, retain=False):
    """Values [batch, n_ops].  With `retain`, the final states stay in the workspace for one
    `expectation_vjp_retained` (same bits and params); `self.retained` is then a token, or None
    when the batch was too large to keep."""
    bits, params = self._prep(bits, params)
    out = torch.empty((bits.shape[0], self.n_ops), dtype=torch.float32,
                      device=self.device)
    fn = self._lib.qhbm_expectation_retain if retain else self._lib.qhbm_expectation
    self.retained = None
    with torch.cuda.device(self.device):
      self._check(fn(self._h, bits.data_ptr(), bits.shape[0], params.data_ptr(), out.data_ptr(),
                     self._stream()))
    # (the C side keeps nothing when the batch exceeds one backward chunk: ask, do not assume)
    if retain and bits.shape[0] > 0 and self.retained_states() == bits.shape[0]:
      self._retain_count = getattr(self, "_retain_count", 0) + 1
      self.retained = self._retain_count
    return out

  def expectation_vjp_retained(self, bits, params, upstream):
    """grad[n_params] from the states kept by `expectation(..., retain=True)`; raises
    EngineError if they are gone (then call `expectation_vjp`)."""
    bits, params = self._prep(bits, params)
    upstream = torch.as_tensor(upstream).to(
        device=self.device, dtype=torch.float32).contiguous()
    if tuple(upstream.shape) != (bits.shape[0], self.n_ops):
      raise ValueError("upstream must have shape [batch, n_ops]")
    grad = torch.zeros((self.n_params,), dtype=torch.float32, device=self.device)
    self.retained = None
    with torch.cuda.device(self.device):
      self._check(
          self._lib.qhbm_expectation_vjp_retained(self._h, bits.data_ptr(), bits.shape[0],
                                                  params.data_ptr(), upstream.data_ptr(),
                                                  grad.data_ptr(), self._stream()))
    return grad

  def expectation_vjp(self, bits, params, upstream, method=GRAD_ADJOINT):
    self.retained = None
    bits, params = self._prep(bits, params)
    upstream = torch.as_tensor(upstream).to(
        device=self.device, dtype=torch.float32).contiguous()
    if tuple(upstream.shape) != (bits.shape[0], self.n_ops):
      raise ValueError("upstream must have shape [batch, n_ops]")
    vals = torch.empty((bits.shape[0], self.n_ops), dtype=torch.float32,
                       device=self.device)
    grad = torch.zeros((self.n_params,), dtype=torch.float32,
                       device=self.device)
    with torch.cuda.device(self.device):
      self._check(
          self._lib.qhbm_expectation_vjp(self._h, bits.data_ptr(),
                                         bits.shape[0], params.data_ptr(),
                                         upstream.data_ptr(), vals.data_ptr(),
                                         grad.data_ptr(), int(method),
                                         self._stream()))
    return vals, grad

  def state_gradients(self, num_states):
    """[num_states, n_params] rows of the last adjoint VJP (their sum over states is its gradient)."""
    rows = torch.empty((int(num_states), self.n_params), dtype=torch.float32, device=self.device)
    with torch.cuda.device(self.device):
      self._check(self._lib.qhbm_state_gradients(self._h, int(num_states), rows.data_ptr(), self._stream()))
    return rows

  def statevector(self, bits, params):
    """Final states C(params)|x_u>, complex64 [batch, 2^n] (qubit 0 = most significant bit)."""
    self.retained = None
    bits, params = self._prep(bits, params)
    out = torch.empty((bits.shape[0], 1 << self.n_qubits), dtype=torch.complex64,
                      device=self.device)
    with torch.cuda.device(self.device):
      self._check(
          self._lib.qhbm_statevector(self._h, bits.data_ptr(), bits.shape[0],
                                     params.data_ptr(), out.data_ptr(),
                                     self._stream()))
    return out

  def sample(self, bits, params, n_shots, seed=0, shift_gate=-1, shift=0.0):
    """int8 [batch, n_shots, n_qubits]: computational-basis samples of C(params)|x_u>;
    `shift_gate`/`shift` select one parameter-shifted program (see include/qhbm_engine.h)."""
    self.retained = None
    bits, params = self._prep(bits, params)
    out = torch.empty((bits.shape[0], int(n_shots), self.n_qubits), dtype=torch.int8,
                      device=self.device)
    with torch.cuda.device(self.device):
      self._check(
          self._lib.qhbm_sample(self._h, bits.data_ptr(), bits.shape[0], params.data_ptr(),
                                int(n_shots), int(seed) & (2**64 - 1), int(shift_gate),
                                float(shift), out.data_ptr(), self._stream()))
    return out

  def sample_counts(self, bits, params, n_shots, seed=0, shift_gates=(-1,), shifts=(0.0,)):
    """int32 [n_programs, batch, 2^n]: how many of `n_shots` shots of every (shifted program, state)
    pair gave each outcome -- all programs in one launch set (include/qhbm_engine.h qhbm_sample_counts)."""
    self.retained = None
    bits, params = self._prep(bits, params)
    sg = np.ascontiguousarray(shift_gates, dtype=np.int32)
    sv = np.ascontiguousarray(shifts, dtype=np.float32)
    if sg.shape != sv.shape or sg.ndim != 1:
      raise ValueError("shift_gates and shifts must be 1-D and of equal length")
    out = torch.empty((len(sg), bits.shape[0], 1 << self.n_qubits), dtype=torch.int32, device=self.device)
    with torch.cuda.device(self.device):
      self._check(
          self._lib.qhbm_sample_counts(self._h, bits.data_ptr(), bits.shape[0], params.data_ptr(), len(sg),
                                       sg.ctypes.data, sv.ctypes.data, int(n_shots), int(seed) & (2**64 - 1),
                                       out.data_ptr(), self._stream()))
    return out

  def expectation_jacobian(self, bits, params):
    self.retained = None
    bits, params = self._prep(bits, params)
    vals = torch.empty((bits.shape[0], self.n_ops), dtype=torch.float32,
                       device=self.device)
    jac = torch.zeros((bits.shape[0], self.n_ops, self.n_params),
                      dtype=torch.float32, device=self.device)
    with torch.cuda.device(self.device):
      self._check(
          self._lib.qhbm_expectation_jacobian(self._h, bits.data_ptr(),
                                              bits.shape[0], params.data_ptr(),
                                              vals.data_ptr(), jac.data_ptr(),
                                              self._stream()))
    return vals, jac
