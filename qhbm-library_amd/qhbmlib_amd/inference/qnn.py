"""Inference on quantum circuits (reference: qhbmlib/inference/qnn.py).

`AnalyticQuantumInference` is the drop-in for the reference class of the same
name: where the reference tiles protos and calls `tfq.layers.Expectation`
(qnn.py:112,133-138), this one hands the unique bitstrings and the current
symbol values to the HIP engine through the C ABI.  There is no CPU path.
"""
import abc
import collections
import math
from typing import Sequence, Union

import torch
import torch.distributed as dist

from qhbmlib_amd import _engine
from qhbmlib_amd import ir
from qhbmlib_amd import parallel
from qhbmlib_amd import utils
from qhbmlib_amd.inference import ebm
from qhbmlib_amd.models import circuit  # noqa: F401
from qhbmlib_amd.models import energy
from qhbmlib_amd.models import hamiltonian

Observables = Union[Sequence[ir.PauliSumLike], hamiltonian.Hamiltonian]


class _ExpectationFunction(torch.autograd.Function):
  """values[U, T] = engine(bits, symbol_values); backward = adjoint VJP in the engine
  (the role TFQ's adjoint differentiator plays at qnn.py:90-99).

  With a process group the unique rows are dealt out in contiguous blocks
  (`parallel.partition`): every rank simulates its block, the values are all-gathered and the
  [P] gradient is all-reduced, so `vqt()` / `qmhl()` scale over the GPUs of a node without user
  code.  Every rank must pass the same bitstrings and parameters;
  `AnalyticQuantumInference._expectation` verifies that before the call (`check_consistency`)."""

  @staticmethod
  def forward(ctx, symbol_values, engine, bits, method, group, ordered, grad_mask=None, weights=None):
    ctx.engine, ctx.method, ctx.group, ctx.ordered = engine, method, group, ordered
    ctx.rows = None
    # symbols nobody wants a gradient of (a fixed data circuit's): no gradient work, and the backward sweep stops
    # at the first gate that is not theirs (qhbm_set_gradient_mask; a no-op unless the mask changed)
    ctx.grad_mask = None if grad_mask is None or all(grad_mask) else tuple(bool(f) for f in grad_mask)
    if ctx.needs_input_grad[0]:
      engine.set_gradient_mask(ctx.grad_mask)
    if group is not None:
      blocks = parallel.partition(bits.shape[0], dist.get_world_size(group), weights)
      ctx.rows = blocks[dist.get_rank(group)]
      ctx.blocks = blocks
      bits = bits[ctx.rows[0]:ctx.rows[1]]
    ctx.bits = bits
    ctx.save_for_backward(symbol_values)
    # the forward leaves its final states in the engine's workspace: if nothing else runs on this
    # engine before backward(), the backward sweep starts from them instead of simulating again
    retain = method == _engine.GRAD_ADJOINT and ctx.needs_input_grad[0]
    out = engine.expectation(bits, symbol_values.detach(), retain=retain)
    ctx.token = engine.retained
    if group is not None:
      out = parallel.all_gather_rows(out, ctx.blocks, group)
    return out

  @staticmethod
  def backward(ctx, upstream):
    (symbol_values,) = ctx.saved_tensors
    eng = ctx.engine
    eng.set_gradient_mask(ctx.grad_mask)  # (another forward on this engine may have set another one: then the plans are rebuilt)
    upstream = upstream.contiguous()
    if ctx.rows is not None:
      upstream = upstream[ctx.rows[0]:ctx.rows[1]].contiguous()
    grad = None
    if ctx.token is not None and eng.retained == ctx.token:
      try:
        grad = eng.expectation_vjp_retained(ctx.bits, symbol_values.detach(), upstream)
      except _engine.EngineError:
        grad = None  # the states are gone after all: simulate again
    if grad is None:
      _, grad = eng.expectation_vjp(ctx.bits, symbol_values.detach(), upstream, ctx.method)
    if ctx.group is not None:
      if ctx.ordered and ctx.method == _engine.GRAD_ADJOINT:
        # every rank's per-state rows, gathered in global state order and added in that order: the
        # same [U, P] array and the same reduction for ANY number of ranks -> bit-identical
        rows = eng.state_gradients(ctx.bits.shape[0]) if ctx.bits.shape[0] else grad.new_zeros((0, grad.numel()))
        rows = parallel.all_gather_rows(rows, ctx.blocks, ctx.group)
        grad = rows.to(torch.float64).sum(0).to(torch.float32)
      else:
        parallel.all_reduce_sum(grad, ctx.group)
    return grad.to(symbol_values.device), None, None, None, None, None, None, None


class ResolvedCircuits(tuple):
  """What `QuantumCircuit.__call__(bitstrings)` hands to `_expectation` where the reference hands
  `[U]` serialized protos (circuit.py:129-136): the unique bitstrings and the circuit they
  initialise.  Unpacks as `(bitstrings, circuit)`."""

  gradient_mask = None  # optional: one bool per symbol, False = nobody wants that gradient (set by `expectation`)

  def __new__(cls, bitstrings, circuit):  # pylint: disable=redefined-outer-name
    return super().__new__(cls, (bitstrings, circuit))

  @property
  def bitstrings(self):
    return self[0]

  @property
  def circuit(self):
    return self[1]

  @property
  def num_circuits(self):
    return int(self[0].shape[0])


class QuantumInference(torch.nn.Module, abc.ABC):
  """Interface for inference on quantum circuits (qnn.py:29-84)."""

  def __init__(self, input_circuit: circuit.QuantumCircuit, name: Union[None, str] = None):
    super().__init__()
    self.name = name or type(self).__name__
    input_circuit.build([])
    self._circuit = input_circuit

  @property
  def circuit(self):
    return self._circuit

  def expectation(self, initial_states: torch.Tensor, observables: Observables):
    """[batch_size, n_ops] of <x_i| C^dagger O_j C |x_i>, rows in input order
    (qnn.py:50-80).  `observables` is a list of PauliSums or a Hamiltonian.

    Same steps as the reference: dedup, total circuit, `circuits = total_circuit(unique_states)`,
    symbol values tiled to [num_circuits, P] (here a stride-0 view, no copy), the abstract
    `_expectation(circuits, symbol_names, symbol_values, observables)`, expand."""
    initial_states = torch.as_tensor(initial_states)
    unique_states, idx, _ = utils.unique_bitstrings_with_counts(initial_states)
    if isinstance(observables, hamiltonian.Hamiltonian):
      total_circuit = self.circuit + observables.circuit_dagger
    else:
      total_circuit = self.circuit
    circuits = ResolvedCircuits(*total_circuit(unique_states))
    values, flags = total_circuit.symbol_values_and_flags()   # (one pass through the value layers)
    tiled_values = values.unsqueeze(0).expand(circuits.num_circuits, -1)
    if tiled_values.requires_grad:  # which symbols a backward pass will want (e.g. not a fixed data circuit's)
      circuits.gradient_mask = flags
    unique_expectations = self._expectation(circuits, total_circuit.symbol_names, tiled_values, observables)
    return utils.expand_unique_results(unique_expectations, idx)

  @abc.abstractmethod
  def _expectation(self, circuits, symbol_names, symbol_values, observables):
    """The plugin point, with the reference's signature (qnn.py:82-84): `circuits` is a
    `ResolvedCircuits` (bitstrings [U, n] int8 + the total circuit), `symbol_names` the [P] names,
    `symbol_values` [U, P] with identical rows, `observables` as given to `expectation`.
    Returns [U, n_ops]."""
    raise NotImplementedError()


def _row_of_tiled(symbol_values: torch.Tensor, total_circuit) -> torch.Tensor:
  """The [P] parameter vector behind the [U, P] tile of qnn.py:75-76 (the engine broadcasts it)."""
  if symbol_values.dim() == 1:
    return symbol_values
  if symbol_values.shape[0] == 0:
    return total_circuit.symbol_values
  if symbol_values.stride(0) != 0 and not bool((symbol_values == symbol_values[:1]).all()):
    raise ValueError("the engine takes one parameter vector per call: rows of symbol_values differ")
  return symbol_values[0]


def _engine_bits(total_circuit, states):
  """Bitstring columns in engine-qubit order."""
  perm = total_circuit.bit_column_to_qubit()
  bits = states.to(torch.int8)
  if perm != list(range(len(perm))):
    permuted = torch.zeros_like(bits)
    permuted[:, perm] = bits
    bits = permuted
  return bits


class _ContentKey:
  """A content key of the engine cache with its hash computed once."""
  __slots__ = ("content", "_hash")

  def __init__(self, content):
    self.content = content
    self._hash = hash(content)

  def __hash__(self):
    return self._hash

  def __eq__(self, other):
    return self is other or (isinstance(other, _ContentKey) and self._hash == other._hash and self.content == other.content)


class _EngineCache:
  """Engines keyed by CONTENT -- (n, flat gate list, n symbols, Pauli masks of every op) -- never
  by object identity: CPython reuses the ids of freed operator lists, and `PauliSum.__iadd__`
  mutates in place.  Least-recently-used engines are dropped when the cache holds more than
  `max_engines` or their workspaces more than `max_bytes`; a dropped engine frees its device
  memory when the last autograd graph that still references it is gone."""

  def __init__(self, max_engines=4, max_bytes=160 << 30):
    self.max_engines, self.max_bytes = max_engines, max_bytes
    self._engines = collections.OrderedDict()

  def __len__(self):
    return len(self._engines)

  def get(self, key, make):
    eng = self._engines.get(key)
    if eng is None:
      eng = make()
      self._engines[key] = eng
    self._engines.move_to_end(key)
    while len(self._engines) > 1 and (
        len(self._engines) > self.max_engines or
        sum(e.allocated_bytes() for e in self._engines.values()) > self.max_bytes):
      self._engines.popitem(last=False)
    return eng


class AnalyticQuantumInference(QuantumInference):
  """Exact expectation values with adjoint gradients on the MI355X engine
  (qnn.py:87-139).  `gradient_method` may be set to
  `_engine.GRAD_PARAMETER_SHIFT` to use two shifted forwards per gate occurrence
  instead (the rule of tfq.differentiators.ParameterShift, qnn.py:168).

  `process_group`: a `torch.distributed` group (or True for the default group) over which the
  unique bitstrings are sharded, one process per GPU; None (default) runs on this process's GPU
  only.  The gradient leaves a sharded call through ONE all-reduce of the [P] vector (default: config 3
  moves 3.7 KiB per step and rank; the sum order, hence the last bits, depend on the number of ranks);
  `ordered_reduction=True` gathers the per-state gradient rows [U, P] instead (config 3: 15.5 MB) and adds
  them in global state order in fp64, so losses and gradients are bit-identical for 1, 2, 4 or 8 ranks --
  for regression runs that compare rank counts, not for throughput.
  `shard_weights`: one positive speed per rank, the same list on every rank (`parallel.measured_weights`, taken once from
  a warm-up step): the unique rows are then dealt out in blocks PROPORTIONAL to them instead of equal ones -- GPUs of one
  node differ by 6-8 % in sustained clock and ranks in lock step run at the slowest one's pace.  Not with
  `ordered_reduction=True`.
  `check_consistency` (default True): before a sharded call the ranks compare a fingerprint of the
  unique bitstrings and the symbol values (8 bytes each) and raise `parallel.ShardMismatchError` if
  they differ -- differently seeded samplers would otherwise shard different sets, silently."""

  MAX_OPS_PER_CALL = 1024  # kMaxOps of the engine (csrc/program.h)

  def __init__(self, input_circuit: circuit.QuantumCircuit, name: Union[None, str] = None,
               device: Union[None, int] = None, gradient_method: int = _engine.GRAD_ADJOINT,
               process_group=None, max_cached_engines: int = 4, ordered_reduction: bool = False,
               check_consistency: bool = True, shard_weights=None):
    super().__init__(input_circuit, name)
    if shard_weights is not None and ordered_reduction:
      raise ValueError("ordered_reduction=True (regression runs that compare rank counts) uses equal blocks only")
    self.shard_weights = None if shard_weights is None else [float(w) for w in shard_weights]
    self._device = device
    self.gradient_method = gradient_method
    self._process_group = process_group
    self.ordered_reduction = ordered_reduction
    self.check_consistency = check_consistency
    self._engines = _EngineCache(max_cached_engines)
    self._by_identity = {}

  def _group(self):
    g = self._process_group
    if g is None or g is False:
      return None
    if not (dist.is_available() and dist.is_initialized()):
      raise _engine.EngineError("process_group given but torch.distributed is not initialised")
    return dist.group.WORLD if g is True else g

  def _engine_for(self, n_qubits, flat_gates, n_symbols, op_masks):
    # `flat_gates` and the mask lists are memoised by the IR (ir.Circuit.flat_gates, ir.PauliSum.masks): while circuit and
    # operators are unchanged the SAME objects arrive, and an identity lookup finds the engine without hashing ~10^4
    # numbers per step; anything else is looked up by CONTENT as before.
    ident = (id(flat_gates), n_qubits, n_symbols) + tuple(map(id, op_masks))
    seen = self._by_identity.get(ident)
    if seen is not None:
      key = seen[0]
    else:
      key = _ContentKey((n_qubits, tuple(flat_gates), n_symbols, tuple(tuple(m) for m in op_masks)))
      if len(self._by_identity) >= 64:
        self._by_identity.clear()
      self._by_identity[ident] = (key, flat_gates, list(op_masks))   # (the references keep the ids valid)

    def make():
      if not torch.cuda.is_available():
        raise _engine.EngineError(
            "AnalyticQuantumInference needs an MI355X: the expectation engine is HIP-only "
            "and has no CPU fallback")
      dev = torch.cuda.current_device() if self._device is None else self._device
      eng = _engine.Engine(dev)
      eng.set_circuit(n_qubits, flat_gates, n_symbols)
      eng.set_observables([list(m) for m in op_masks])
      return eng
    return self._engines.get(key, make)

  def _expectation(self, circuits, symbol_names, symbol_values, observables):
    """See qnn.py:114-139.  A Hamiltonian is only accepted if its energy inherits from
    PauliMixin."""
    if isinstance(observables, hamiltonian.Hamiltonian):
      if not isinstance(observables.energy, energy.PauliMixin):
        raise TypeError("General Hamiltonians not accepted.  "
                        "Please use `SampledQuantumInference` instead.")
      ops = observables.operator_shards
      post_process = lambda y: observables.energy.operator_expectation(y).unsqueeze(-1)
    else:
      ops = list(observables)
      post_process = lambda x: x
    unique_states, total_circuit = circuits
    qubits = total_circuit.qubits
    bits = _engine_bits(total_circuit, unique_states)
    values = _row_of_tiled(symbol_values, total_circuit).to(torch.float32)
    flat_gates = total_circuit.pqc.flat_gates(qubits, list(symbol_names))
    group = self._group()
    if group is not None and self.check_consistency:
      parallel.assert_same_on_all_ranks("the unique bitstrings or the symbol values", bits, values, group=group)
    # one engine call measures at most MAX_OPS_PER_CALL observables (its LDS accumulators);
    # longer lists -- e.g. the 1350 shards of a third-order KOBE on 20 qubits -- go in slices
    parts = []
    for lo in range(0, max(len(ops), 1), self.MAX_OPS_PER_CALL):
      masks = [ir.as_pauli_sum(op).masks(qubits) for op in ops[lo:lo + self.MAX_OPS_PER_CALL]]
      eng = self._engine_for(len(qubits), flat_gates, len(symbol_names), masks)
      grad_mask = getattr(circuits, "gradient_mask", None)
      if grad_mask is not None and len(grad_mask) != len(symbol_names):
        grad_mask = None
      parts.append(_ExpectationFunction.apply(values, eng, bits, self.gradient_method, group, self.ordered_reduction,
                                              grad_mask, self.shard_weights))
    expectations = parts[0] if len(parts) == 1 else torch.cat(parts, 1)
    return post_process(expectations)


class _ParameterShiftSurrogate(torch.autograd.Function):
  """Zero-valued term whose backward is the two-term parameter-shift rule on SAMPLED estimates
  (tfq.differentiators.ParameterShift as driven at qnn.py:188-226): for every gate occurrence
  with exponent c*s + o, d/ds = (pi c / 2) [f(o + 1/2) - f(o - 1/2)].  `estimator(shift_gates, shifts)`
  returns the no-grad estimates [Q, U, T] of Q shifted programs at once: all 2 G programs of a backward
  run as ONE batched engine call per measurement basis (`qhbm_sample_counts`), not 2 G forwards."""

  @staticmethod
  def forward(ctx, symbol_values, estimator, gates, shape):
    ctx.estimator, ctx.gates = estimator, gates
    ctx.save_for_backward(symbol_values)
    return torch.zeros(shape, dtype=torch.float32, device=symbol_values.device)

  @staticmethod
  def backward(ctx, upstream):
    (symbol_values,) = ctx.saved_tensors
    grad = torch.zeros_like(symbol_values)
    shifted = []
    for g, gate in enumerate(ctx.gates):
      kind, pidx, scalar = gate[0], gate[3], gate[4]
      if pidx < 0 or kind == _engine.GATE_I:
        continue
      if kind == _engine.GATE_ISWAPPOW:
        raise _engine.EngineError("the two-term parameter-shift rule does not apply to ISWAPPOW")
      shifted.append((g, pidx, scalar))
    if not shifted:
      return grad, None, None, None
    gates = [g for g, _, _ in shifted for _ in (0, 1)]
    shifts = [sh for _ in shifted for sh in (0.5, -0.5)]
    est = ctx.estimator(gates, shifts)                                   # [2 G, U, T]
    diff = est[0::2] - est[1::2]                                         # f(+1/2) - f(-1/2) per gate
    per_gate = torch.sum(upstream.to(diff.device).unsqueeze(0) * diff, dim=(1, 2))   # [G]
    weights = torch.tensor([0.5 * math.pi * scalar for _, _, scalar in shifted], dtype=per_gate.dtype,
                           device=per_gate.device)
    index = torch.tensor([pidx for _, pidx, _ in shifted], dtype=torch.long, device=grad.device)
    grad.index_add_(0, index, (weights * per_gate).to(grad.device, grad.dtype))
    return grad, None, None, None


class SampledQuantumInference(QuantumInference):
  """Sampling methods for inference on QuantumCircuit objects (qnn.py:142-292): expectation
  values are averages over `expectation_samples` computational-basis shots drawn by the engine,
  derivatives use the parameter-shift rule on sampled estimates.  Estimates are taken from per-outcome
  shot COUNTS (`qhbm_sample_counts`: every shifted program of a backward pass in one launch set) up to
  `MAX_COUNT_QUBITS` qubits, from materialised shots (`qhbm_sample`, one program per call) above."""

  MAX_COUNT_QUBITS = 16   # 2^n counters per (program, state)
  # The counts of one engine call are [programs, states, 2^n] int32 and their frequencies the same again in float32:
  # a backward pass asks for 2 G programs at once, which at 16 qubits, ~100 gates and ~100 states would be tens of
  # GB.  The programs therefore go in slices of at most this many bytes, each reduced to its [Q, U, T] estimates
  # before the next is drawn; a single program above the budget falls back to materialised shots.
  COUNT_BYTES_BUDGET = 1 << 30

  def _count_slice(self, n_programs, n_states, n_qubits):
    """Programs per `qhbm_sample_counts` call (0: not even one fits -- use shots)."""
    per_program = max(1, n_states) * (1 << n_qubits) * 8
    if n_qubits > self.MAX_COUNT_QUBITS or per_program > self.COUNT_BYTES_BUDGET:
      return 0
    return max(1, min(n_programs, self.COUNT_BYTES_BUDGET // per_program))

  def __init__(self, input_circuit: circuit.QuantumCircuit, expectation_samples: int,
               name: Union[None, str] = None, device: Union[None, int] = None,
               initial_seed: Union[None, int] = None):
    super().__init__(input_circuit, name)
    self._expectation_samples = int(expectation_samples)
    self._device = device
    self._engines = {}
    self._seed = int(ebm.fresh_seed() if initial_seed is None else initial_seed) & (2**63 - 1)

  def _next_seed(self):
    self._seed = (self._seed * 6364136223846793005 + 1442695040888963407) & (2**63 - 1)
    return self._seed

  def _engine_for(self, qubits, flat_gates, n_symbols):
    key = (len(qubits), tuple(flat_gates), n_symbols)
    eng = self._engines.get(key)
    if eng is None:
      if not torch.cuda.is_available():
        raise _engine.EngineError(
            "SampledQuantumInference needs an MI355X: the engine is HIP-only, no CPU fallback")
      dev = torch.cuda.current_device() if self._device is None else self._device
      eng = _engine.Engine(dev)
      eng.set_circuit(len(qubits), flat_gates, n_symbols)
      self._engines[key] = eng
    return eng

  _engine_bits = staticmethod(_engine_bits)

  def _counts(self, eng, bits, values, shift_gates, shifts):
    """float32 [Q, U, 2^n] shot frequencies of Q shifted programs (one engine call)."""
    counts = eng.sample_counts(bits, values, self._expectation_samples, self._next_seed(), shift_gates, shifts)
    return counts.to(torch.float32) / float(self._expectation_samples)

  def _pauli_estimator(self, total_circuit, bits, values, strings):
    """estimator(shift_gates, shifts) -> [Q, U, len(strings)] sampled <P> of each Pauli string under each
    of the Q shifted programs.  Strings are grouped by their X/Y pattern; a group shares one rotated
    circuit (X: H, Y: rx(pi/2), appended after the circuit) and one batch of shots."""
    qubits = total_circuit.qubits
    names = total_circuit.symbol_names
    n = len(qubits)
    base = total_circuit.pqc.flat_gates(qubits, names)
    qindex = {q: i for i, q in enumerate(qubits)}
    groups = {}
    for k, st in enumerate(strings):
      key = tuple(sorted((qindex[q], p) for q, p in st.paulis.items() if p in ("X", "Y")))
      groups.setdefault(key, []).append(k)
    use_counts = n <= self.MAX_COUNT_QUBITS
    outcomes = torch.arange(1 << n, dtype=torch.int64) if use_counts else None
    plans = []
    for key, members in groups.items():
      rot = ir.Circuit([ir.H(qubits[i]) if p == "X" else ir.rx(math.pi / 2)(qubits[i]) for i, p in key])
      eng = self._engine_for(qubits, base + rot.flat_gates(qubits, names), len(names))
      masks = torch.zeros((len(members), n), dtype=torch.float32)
      for row, k in enumerate(members):
        for q in strings[k].paulis:
          masks[row, qindex[q]] = 1.0
      signs = None
      if use_counts:   # signs[x, m] = (-1)^{popc(x & string m)}; qubit i is bit n - 1 - i of the outcome index
        weights = (masks.to(torch.int64) * (1 << torch.arange(n - 1, -1, -1, dtype=torch.int64))).sum(1)   # [members]
        par = outcomes.unsqueeze(1) & weights.unsqueeze(0)
        par = par ^ (par >> 32)
        for sh in (16, 8, 4, 2, 1):
          par = par ^ (par >> sh)
        signs = (1.0 - 2.0 * (par & 1).to(torch.float32)).to(bits.device if bits.is_cuda else eng.device)
      plans.append((eng, members, masks, signs))

    def estimator(shift_gates=(-1,), shifts=(0.0,)):
      shift_gates, shifts = list(shift_gates), list(shifts)
      out = torch.ones((len(shift_gates), bits.shape[0], len(strings)), dtype=torch.float32, device=values.device)
      step = self._count_slice(len(shift_gates), bits.shape[0], n) if use_counts else 0
      for eng, members, masks, signs in plans:
        if signs is not None and step:
          for lo in range(0, len(shift_gates), step):                               # bounded slices of programs
            freq = self._counts(eng, bits, values, shift_gates[lo:lo + step], shifts[lo:lo + step])   # [q, U, 2^n]
            out[lo:lo + step, :, members] = torch.matmul(freq, signs.to(freq.device)).to(out.device)
            del freq
        else:                                                                       # wide registers: shots, one program at a time
          for q, (g, sh) in enumerate(zip(shift_gates, shifts)):
            shots = eng.sample(bits, values, self._expectation_samples, self._next_seed(), g, sh)
            ones = torch.matmul(shots.to(torch.float32), masks.to(shots.device).t())   # [U, shots, members]
            parity = 1.0 - 2.0 * torch.remainder(ones, 2.0)
            out[q][:, members] = parity.mean(1).to(out.device)
      return out
    return estimator, base

  def _expectation(self, circuits, symbol_names, symbol_values, observables):
    """qnn.py:228-264.  Pauli-sum observables and PauliMixin Hamiltonians are estimated term by
    term; any other Hamiltonian averages `energy(x)` over shots of circuit + hamiltonian.circuit_dagger
    (`_sampled_expectation`, qnn.py:170-226)."""
    unique_states, total_circuit = circuits
    del symbol_names  # the circuit's own order
    bits = self._engine_bits(total_circuit, unique_states)
    symbol_values = _row_of_tiled(symbol_values, total_circuit).to(torch.float32)
    values = symbol_values.detach()
    if isinstance(observables, hamiltonian.Hamiltonian) and not isinstance(observables.energy, energy.PauliMixin):
      qubits, names = total_circuit.qubits, total_circuit.symbol_names
      n = len(qubits)
      base = total_circuit.pqc.flat_gates(qubits, names)
      eng = self._engine_for(qubits, base, len(names))
      energy_device = next(iter(observables.energy.parameters()), torch.zeros(())).device

      def weighted_energy(keys, weights, n_rows):
        """[n_rows, 1]: sum over the DISTINCT (row, bitstring) pairs `keys` = row << n | outcome of
        energy(bitstring) * weight -- the energy is evaluated once per pair, on its own device."""
        rows = ((keys.unsqueeze(1) >> torch.arange(n - 1, -1, -1, device=keys.device)) & 1).to(torch.int8)
        e = observables.energy(rows.to(energy_device))
        out = torch.zeros(n_rows, dtype=e.dtype, device=e.device)
        return out.index_add(0, (keys >> n).to(e.device), e * weights.to(e.device, e.dtype)).unsqueeze(1)

      def mean_energy(shift_gates, shifts):
        """[Q, U, 1] shot average of energy(x) under each of the Q shifted programs."""
        n_prog, n_states = len(shift_gates), bits.shape[0]
        step = self._count_slice(n_prog, n_states, n)
        if step:
          parts = []
          for lo in range(0, n_prog, step):                               # bounded slices of programs
            q = min(step, n_prog - lo)
            freq = self._counts(eng, bits, values, shift_gates[lo:lo + q], shifts[lo:lo + q]).reshape(q * n_states, 1 << n)
            nz = torch.nonzero(freq)                                      # the outcomes that occurred
            keys = (nz[:, 0] << n) | nz[:, 1]
            parts.append(weighted_energy(keys, freq[nz[:, 0], nz[:, 1]], q * n_states).reshape(q, n_states, 1))
            del freq
          return parts[0] if len(parts) == 1 else torch.cat(parts, 0)
        outs = []
        for g, sh in zip(shift_gates, shifts):
          samples = eng.sample(bits, values, self._expectation_samples, self._next_seed(), g, sh)
          place = (1 << torch.arange(n - 1, -1, -1, device=samples.device, dtype=torch.int64))
          key = (samples.to(torch.int64) * place).sum(-1) + (
              torch.arange(n_states, device=samples.device, dtype=torch.int64).unsqueeze(1) << n)
          uniq, counts = torch.unique(key, return_counts=True)
          outs.append(weighted_energy(uniq, counts.to(torch.float32) / float(samples.shape[1]), n_states))
        return torch.stack(outs)

      def estimator(shift_gates, shifts):
        with torch.no_grad():
          return mean_energy(list(shift_gates), list(shifts)).to(torch.float32)

      forward_pass = mean_energy([-1], [0.0])[0]  # differentiable w.r.t. the energy's variables
      surrogate = _ParameterShiftSurrogate.apply(symbol_values, estimator, base, tuple(forward_pass.shape))
      return forward_pass + surrogate.to(forward_pass.device)

    if isinstance(observables, hamiltonian.Hamiltonian):
      ops = [ir.as_pauli_sum(op) for op in observables.operator_shards]
      post_process = lambda y: observables.energy.operator_expectation(y).unsqueeze(-1)
    else:
      ops = [ir.as_pauli_sum(op) for op in observables]
      post_process = lambda x: x
    strings, coeffs = [], []
    for t, op in enumerate(ops):
      for term in op.terms:
        strings.append(term)
        coeffs.append((len(strings) - 1, t, term.coefficient))
    mix = torch.zeros((len(strings), len(ops)), dtype=torch.float32)
    for k, t, c in coeffs:
      mix[k, t] = c
    estimator, base = self._pauli_estimator(total_circuit, bits, values, strings)
    with torch.no_grad():
      estimates = estimator()[0]
    surrogate = _ParameterShiftSurrogate.apply(symbol_values, estimator, base, tuple(estimates.shape))
    return post_process(torch.matmul(estimates + surrogate.to(estimates.device), mix.to(estimates.device)))

  def _sample(self, initial_states: torch.Tensor, counts: torch.Tensor):
    """`ragged[i]` = `counts[i]` bitstrings drawn from circuit|initial_states[i]> (qnn.py:266-292).
    Shots are i.i.d., so the first counts[i] of max(counts) drawn stand for the reference's
    shuffled mask."""
    initial_states = torch.as_tensor(initial_states)
    counts = [int(c) for c in torch.as_tensor(counts).reshape(-1)]
    qubits, names = self.circuit.qubits, self.circuit.symbol_names
    eng = self._engine_for(qubits, self.circuit.pqc.flat_gates(qubits, names), len(names))
    bits = self._engine_bits(self.circuit, initial_states)
    values = self.circuit.symbol_values.detach().to(torch.float32)
    shots = eng.sample(bits, values, max(counts) if counts else 0, self._next_seed())
    return [shots[i, :c] for i, c in enumerate(counts)]
