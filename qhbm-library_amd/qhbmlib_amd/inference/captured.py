"""Whole-step hipGraph replay of a QHBM loss (no counterpart in the reference: its step is a `tf.function`).

At the reference's own sizes (tests <= 5 qubits, the default experiment 4 qubits, BASELINE configs[0..1]) a VQT / QMHL step is
LAUNCH-bound: config 2's engine work is 0.28 ms in about ten kernels, while the eager mirror around it (value layers, the
EBM's parity kernel, weighted averages, autograd bookkeeping) issues dozens of small torch kernels from Python.
`CapturedLoss` records ONE step -- `loss_fn()` and its `backward()` -- into a hipGraph and replays it:

  * the sampler stays OUTSIDE the graph: every step's samples are deduplicated on the device
    (`utils.unique_bitstrings_with_counts`) and written, padded to a fixed capacity with zero-count copies of the first
    row, into static buffers; inside the graph the EBM side averages over that multiset
    (`EnergyInferenceBase.fixed_samples`), so shapes never change and nothing synchronises;
  * the engine's compute calls are asynchronous on the caller's stream and capturable: the
    retained forward and the adjoint sweep from the retained states land in the same graph;
  * gradients arrive in the `.grad` of the given variables (static tensors, rewritten by every replay -- the
    whole-network capture idiom of `torch.cuda.graph`).

The replayed step runs the same kernels on the same inputs as the eager step over the same padded multiset, so it returns
the same bits (tests/test_captured_gpu.py); against the unpadded eager step it differs only by the order of the sample
averages (zero-weight rows).
"""
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch

from qhbmlib_amd import _engine
from qhbmlib_amd import utils
from qhbmlib_amd.inference import ebm


class CapturedLoss:
  """`step = CapturedLoss(lambda: vqt(qhbm, [H], beta), [qhbm.e_inference], variables)`; `loss = step()` draws this
  step's samples, replays the graph and returns the loss (a static tensor; `.grad` of every variable is set).

  `e_inferences`: the `EnergyInference` objects whose sample averages the loss takes (their `num_expectation_samples`
  is the capacity of their multiset buffer).  `variables`: the leaves to differentiate (all on one CUDA device; an
  optimiser may update them in place between steps).  `exact_inferences`: inferences the loss asks only for exact
  quantities of (the model's `log_partition` in `qmhl`): they run `device_only`.  `step(multisets=[(bitstrings, counts), ...])` runs on given
  multisets instead of drawing samples (one per inference, rows <= capacity)."""

  def __init__(self, loss_fn: Callable[[], torch.Tensor], e_inferences: Sequence["ebm.EnergyInference"],
               variables: Sequence[torch.Tensor], warmup: int = 2,
               exact_inferences: Sequence["ebm.EnergyInferenceBase"] = ()):
    self._loss_fn = loss_fn
    self._inferences = list(e_inferences)
    # inferences the loss only asks for exact, device-side quantities (qmhl's log Z of the model): `device_only`
    self._exact = [inf for inf in exact_inferences if all(inf is not other for other in self._inferences)]
    self._variables = [v for v in variables]
    if not self._variables:
      raise ValueError("CapturedLoss needs at least one variable to differentiate")
    devices = {v.device for v in self._variables}
    if len(devices) != 1 or next(iter(devices)).type != "cuda":
      raise _engine.EngineError(
          "CapturedLoss records device work only: move every variable to ONE CUDA device first "
          f"(found {sorted(str(d) for d in devices)}); a host-resident parameter would be copied inside the graph")
    self.device = next(iter(devices))
    self._warmup = int(warmup)
    self._buffers: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
    for inf in self._inferences:
      cap, n = int(inf.num_expectation_samples), int(inf.energy.num_bits)
      self._buffers[id(inf)] = (torch.zeros((cap, n), dtype=torch.int8, device=self.device),
                                torch.zeros((cap,), dtype=torch.int32, device=self.device))
    self._graph: Optional[torch.cuda.CUDAGraph] = None
    self._loss: Optional[torch.Tensor] = None
    self.last_unique_rows = []

  # ---- the multisets ---------------------------------------------------------------------------------
  def _fill(self, inf, multiset):
    bits_buf, counts_buf = self._buffers[id(inf)]
    rows, counts = multiset
    rows = torch.as_tensor(rows).to(device=self.device, dtype=torch.int8)
    counts = torch.as_tensor(counts).to(device=self.device, dtype=torch.int32)
    u, cap = int(rows.shape[0]), int(bits_buf.shape[0])
    if u == 0 or u > cap:
      raise ValueError(f"a multiset of {u} rows for a buffer of {cap} (1 <= rows <= num_expectation_samples)")
    bits_buf[:u].copy_(rows)
    if u < cap:
      bits_buf[u:].copy_(rows[:1].expand(cap - u, -1))   # zero-count copies of the first row: static shape, no new state kinds
    counts_buf[:u].copy_(counts)
    counts_buf[u:].zero_()
    return u

  def _draw(self, inf):
    with torch.no_grad():
      samples = inf.sample(inf.num_expectation_samples)   # the sampler: eager, outside the graph
    rows, _, counts = utils.unique_bitstrings_with_counts(samples.to(self.device))
    return rows, counts

  # ---- one step ----------------------------------------------------------------------------------------
  def _run(self):
    """loss_fn() + backward() over the static multisets (eager or under capture: the same code)."""
    stack = []
    try:
      for inf in self._inferences:
        cm = inf.fixed_samples(*self._buffers[id(inf)])
        cm.__enter__()
        stack.append(cm)
      for inf in self._exact:
        cm = inf.device_only()
        cm.__enter__()
        stack.append(cm)
      loss = self._loss_fn()
      loss.backward()
    finally:
      for cm in reversed(stack):
        cm.__exit__(None, None, None)
    return loss

  def eager(self, multisets=None):
    """The same padded step WITHOUT the graph (what the replay is compared with bit for bit)."""
    self._prepare(multisets)
    for v in self._variables:
      v.grad = None
    return self._run().detach()

  def _prepare(self, multisets):
    if multisets is None:
      multisets = [self._draw(inf) for inf in self._inferences]
    if len(multisets) != len(self._inferences):
      raise ValueError("one (bitstrings, counts) pair per energy inference")
    self.last_unique_rows = [self._fill(inf, ms) for inf, ms in zip(self._inferences, multisets)]

  def _capture(self):
    side = torch.cuda.Stream(device=self.device)
    side.wait_stream(torch.cuda.current_stream(self.device))
    with torch.cuda.stream(side):           # warm-up on a side stream: allocations, plans, engine workspaces
      for _ in range(max(1, self._warmup)):
        for v in self._variables:
          v.grad = None
        self._run()
    torch.cuda.current_stream(self.device).wait_stream(side)
    torch.cuda.synchronize(self.device)
    for v in self._variables:
      v.grad = None                         # backward() inside the capture allocates the static .grad tensors
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
      loss = self._run()
    self._graph, self._loss = graph, loss.detach()
    self._grads = [v.grad for v in self._variables]

  def __call__(self, multisets=None):
    self._prepare(multisets)
    if self._graph is None:
      with torch.cuda.device(self.device):
        self._capture()
    for v, g in zip(self._variables, self._grads):
      if g is not None and v.grad is not g:                   # (an optimiser's zero_grad(set_to_none=True) drops them: put the static ones back)
        v.grad = g
    self._graph.replay()
    return self._loss

  @property
  def captured(self):
    return self._graph is not None
