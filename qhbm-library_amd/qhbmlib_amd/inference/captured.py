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
  * gradients are taken with `torch.autograd.grad` inside the graph and copied into static buffers, which every call
    installs as the `.grad` of the given variables (rewritten by every replay).

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
               exact_inferences: Sequence["ebm.EnergyInferenceBase"] = (), synchronize: bool = True):
    self._loss_fn = loss_fn
    # `synchronize` (default): a call returns when its replay has FINISHED.  torch records memset nodes of its own into
    # the graph (the semaphores of a multi-block reduction: `logsumexp` over the 2^20 energies of an analytic EBM), and on
    # this runtime a graph whose replays interleave with other work on the caller's stream that the host never waits for
    # -- even one unrelated element-wise kernel per step -- came back with a wrong log-partition value after about ten
    # steps, and stayed wrong (the engine's own outputs inside the same graph were right; round 6, HISTORY.md).  Waiting
    # for the caller's stream after every replay removes the overlap; a step this class is meant for is launch-bound, so
    # the wait costs microseconds.
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
    self.synchronize = bool(synchronize)
    self._buffers: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
    for inf in self._inferences:
      cap, n = int(inf.num_expectation_samples), int(inf.energy.num_bits)
      self._buffers[id(inf)] = (torch.zeros((cap, n), dtype=torch.int8, device=self.device),
                                torch.zeros((cap,), dtype=torch.int32, device=self.device))
    self._graph: Optional[torch.cuda.CUDAGraph] = None
    self._loss: Optional[torch.Tensor] = None
    self.last_unique_rows = []
    # The gradients are taken with torch.autograd.grad and COPIED into these buffers by kernels of the recorded stream.
    # (`loss.backward()` would hand them to the leaves' AccumulateGrad nodes, which run on the stream each node was first
    # created on: a cross-stream edge inside the graph -- on this runtime the `.grad` of the circuit parameters came back
    # wrong from the second replay on at 18 qubits and more, while the engine's own output was right.)
    self._static_grads = [torch.zeros_like(v) for v in self._variables]

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
      grads = torch.autograd.grad(loss, self._variables, allow_unused=True)
      for buf, g in zip(self._static_grads, grads):
        # (element-wise KERNELS on purpose: `copy_` of a contiguous tensor is a device-to-device memcpy, which a capture
        # records as a memcpy NODE -- and memcpy / memset nodes of a graph warmed up on one stream and replayed on
        # another were dropped from the second replay on by this runtime; see kernels.hip launch_zero_fill)
        if g is None:
          buf.fill_(0.0)
        else:
          torch.mul(g, 1.0, out=buf)
    finally:
      for cm in reversed(stack):
        cm.__exit__(None, None, None)
    return loss

  def eager(self, multisets=None):
    """The same padded step WITHOUT the graph (what the replay is compared with bit for bit)."""
    self._prepare(multisets)
    loss = self._run().detach()
    self._publish()
    return loss

  def _publish(self):
    for v, buf in zip(self._variables, self._static_grads):
      if v.grad is not buf:
        v.grad = buf

  def _prepare(self, multisets):
    if multisets is None:
      multisets = [self._draw(inf) for inf in self._inferences]
    if len(multisets) != len(self._inferences):
      raise ValueError("one (bitstrings, counts) pair per energy inference")
    self.last_unique_rows = [self._fill(inf, ms) for inf, ms in zip(self._inferences, multisets)]

  def _capture(self):
    # Warm-up AND every replay run on ONE stream, the caller's at the time of recording (`_home`); only the capture itself
    # needs a side stream.  (A graph warmed up on one stream and replayed on another lost its memset / memcpy nodes from
    # the second replay on -- round 6; a caller that later arrives on a different stream is ordered against `_home` with
    # events instead.)
    self._home = torch.cuda.current_stream(self.device)
    for _ in range(max(1, self._warmup)):   # allocations, plans, engine workspaces
      self._run()
    torch.cuda.synchronize(self.device)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
      loss = self._run()
    torch.cuda.synchronize(self.device)
    self._graph, self._loss = graph, loss.detach()

  def __call__(self, multisets=None):
    self._prepare(multisets)
    return self.replay()

  def replay(self):
    """The recorded step on whatever the multiset buffers hold (recording it first if need be)."""
    if self._graph is None:
      with torch.cuda.device(self.device):
        self._capture()
    caller = torch.cuda.current_stream(self.device)
    if caller == self._home:
      self._graph.replay()
    else:
      self._home.wait_stream(caller)          # the multiset buffers and the variables as the caller's stream left them
      with torch.cuda.stream(self._home):
        self._graph.replay()
      caller.wait_stream(self._home)          # loss and gradients are the caller's to read
    if self.synchronize:
      caller.synchronize()
    self._publish()
    return self._loss

  @property
  def captured(self):
    return self._graph is not None
