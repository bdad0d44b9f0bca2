"""Batch sharding of the expectation hot path over the GPUs of one node.

The unit of work is one unique EBM bitstring = one independent statevector that
shares only the parameters (broadcast) and the observable.
Rows are dealt out in contiguous blocks, one process per GPU; a statevector is
never split.  One exchange per call: an all-gather of the per-state expectation
values and an all-reduce (sum) of the [P] gradient -- KiB-sized messages over
RCCL/xGMI (latency-bound), nothing else crosses GPUs.
"""
import hashlib
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


class ShardMismatchError(RuntimeError):
  """The ranks of a process group were handed different inputs for one sharded call."""


def partition(num_rows: int, world_size: int, weights: Optional[Sequence[float]] = None) -> List[Tuple[int, int]]:
  """Contiguous [lo, hi) block per rank.  Without `weights`: equal blocks, the first `num_rows % world_size` ranks get
  one extra row.  With `weights` (one positive number per rank, the SAME list on every rank -- `measured_weights`): block
  sizes proportional to them (largest-remainder rounding, ties to the lower rank), so that ranks whose GPU sustains a
  lower clock -- boxes differ by 6-8 % (profiles/r05_bench_slowbox_c3.json) and ranks in lock step run at the slowest
  one's pace -- get fewer states.  Any contiguous partition gives the same results; only the wall time differs."""
  if weights is None:
    base, extra = divmod(num_rows, world_size)
    sizes = [base + (1 if r < extra else 0) for r in range(world_size)]
  else:
    w = [float(x) for x in weights]
    if len(w) != world_size or any(not (x > 0.0) or x != x or x == float("inf") for x in w):
      raise ValueError(f"partition: {world_size} positive finite weights expected, got {list(weights)!r}")
    total = sum(w)
    exact = [num_rows * x / total for x in w]
    sizes = [int(e) for e in exact]
    # hand out the rows the floors left over, largest fractional part first (ties: lower rank)
    order = sorted(range(world_size), key=lambda r: (-(exact[r] - sizes[r]), r))
    for r in order[:num_rows - sum(sizes)]:
      sizes[r] += 1
  out, lo = [], 0
  for size in sizes:
    out.append((lo, lo + size))
    lo += size
  return out


def measured_weights(seconds_per_state: float, group=None) -> List[float]:
  """Every rank's measured speed (1 / its `seconds_per_state`, e.g. a warm-up step's kernel time divided by its block
  size, or `Engine.clock_probe()["ghz"]` inverted), all-gathered ONCE: the same list on every rank, ready for
  `partition(..., weights=...)`.  COLLECTIVE over `group`.  float64 through the exchange, so every rank rounds alike."""
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
    return [1.0]
  if not seconds_per_state > 0.0:
    raise ValueError("measured_weights: a positive time per state is needed")
  dev = _collective_device(group)
  mine = torch.tensor([1.0 / float(seconds_per_state)], dtype=torch.float64, device=dev)
  every = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
  dist.all_gather(every, mine, group=group)
  return [float(t.cpu().item()) for t in every]


def _via_host(group) -> bool:
  """gloo carries CPU tensors only: CUDA tensors take a host round trip (tests, single-GPU boxes);
  nccl (= RCCL on ROCm) moves device memory over xGMI directly."""
  return dist.get_backend(group) == "gloo"


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
  """In-place sum of `t` over the group."""
  if dist.get_world_size(group) == 1:
    return t
  if t.is_cuda and _via_host(group):
    c = t.cpu()
    dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
    t.copy_(c)
  else:
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
  return t


def all_gather_rows(local: torch.Tensor, blocks: List[Tuple[int, int]], group=None) -> torch.Tensor:
  """Concatenates the ranks' row blocks (`blocks[r]` = rank r's [lo, hi)) into the full tensor,
  identical on every rank.  Blocks are padded to the widest one so a single all-gather serves."""
  world = dist.get_world_size(group)
  if world == 1:   # a group of one rank exchanges nothing (the ordered fp64 row sum that follows stays: it is what makes
    return local   # one rank and N ranks bit-identical)
  rank = dist.get_rank(group)
  lo, hi = blocks[rank]
  width = max(h - l for l, h in blocks)
  host = local.is_cuda and _via_host(group)
  src = local.cpu() if host else local
  padded = torch.zeros((width,) + tuple(local.shape[1:]), dtype=src.dtype, device=src.device)
  padded[:hi - lo] = src
  gathered = [torch.empty_like(padded) for _ in range(world)]
  dist.all_gather(gathered, padded, group=group)
  out = torch.cat([g[:h - l] for g, (l, h) in zip(gathered, blocks)], 0)
  return out.to(local.device) if host else out


def default_group_size() -> int:
  """World size of the default process group, 1 when torch.distributed is not in use."""
  return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _collective_device(group, device=None) -> torch.device:
  """Where a small tensor must live for a collective on `group`: the host for gloo, an explicit GPU for nccl
  (= RCCL) -- `device` if given, else the current one (never an implicit cuda:0 before torch.cuda.set_device)."""
  if _via_host(group):
    return torch.device("cpu")
  if device is not None and torch.device(device).type == "cuda":
    return torch.device(device)
  return torch.device("cuda", torch.cuda.current_device())


def agreed_seed(seed: int, group=None, device=None) -> int:
  """Rank 0's `seed` on every rank of `group` (the default group when None; the identity without
  torch.distributed).  COLLECTIVE: every rank of the group must call it.  The sharded expectation deals ONE set of
  unique bitstrings out over the ranks (reference: one sample set, dedup, then the hot path --
  qhbmlib/inference/ebm.py:271-280), so every rank's sampler must draw the same samples.  Nothing calls this at
  construction time (round 3 did: a hidden collective per `initial_seed=None` object, deadlocking ranks that build
  different objects): `QHBM.agree_seeds()` runs it once, lazily, inside the first sharded call -- which is
  collective anyway -- and `agree_seeds(...)` below is the explicit form."""
  if default_group_size() == 1:
    return int(seed)
  box = torch.tensor([int(seed)], dtype=torch.int64, device=_collective_device(group, device))
  dist.broadcast(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
  return int(box.item())


def agree_seeds(*samplers, group=None) -> None:
  """Gives every sampler (objects with `agree_seed(group)`: the EnergyInference classes) rank 0's seed.
  COLLECTIVE over `group`; call it with the same samplers in the same order on every rank."""
  for s in samplers:
    s.agree_seed(group)


def fingerprint(*arrays) -> int:
  """63-bit content hash (blake2b) of the given tensors / arrays: shapes, dtypes and bytes (host side)."""
  h = hashlib.blake2b(digest_size=8)
  for a in arrays:
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    a = np.ascontiguousarray(a)
    h.update(str((a.shape, a.dtype.str)).encode())
    h.update(a.tobytes())
  return int.from_bytes(h.digest(), "little") & (2**63 - 1)


_GOLDEN = -7046029254386353131  # 0x9E3779B97F4A7C15 as int64


def device_fingerprint(*tensors) -> torch.Tensor:
  """int64 [1] content hash computed WHERE THE TENSORS LIVE (no copy of the data to the host, no sync): every
  tensor's elements, widened to int64 bit patterns, times position-dependent odd multipliers, summed mod 2^64, mixed
  with the shapes.  Not cryptographic -- it only has to tell apart ranks that sampled different bitstrings or hold
  different parameters."""
  dev = next((t.device for t in tensors if torch.is_tensor(t)), torch.device("cpu"))
  acc = torch.zeros((), dtype=torch.int64, device=dev)
  salt = 1
  for t in tensors:
    t = torch.as_tensor(t, device=dev).detach().contiguous()
    if t.dtype in (torch.float32,):
      bits = t.view(torch.int32).to(torch.int64)
    elif t.dtype in (torch.float64,):
      bits = t.view(torch.int64)
    else:
      bits = t.to(torch.int64)
    flat = bits.reshape(-1)
    idx = torch.arange(flat.numel(), dtype=torch.int64, device=dev)
    mult = (idx * _GOLDEN + (2 * salt + 1)) | 1            # int64 arithmetic wraps: that is the modulus
    acc = acc * 31 + ((flat + 1) * mult).sum() + sum((d + 7) * (k + 3) for k, d in enumerate(t.shape)) * salt
    salt += 1
  return (acc & (2**63 - 1)).reshape(1)


def assert_same_on_all_ranks(what: str, *arrays, group=None) -> None:
  """Raises ShardMismatchError (on EVERY rank) unless all ranks of the group hold the same `arrays` (same shapes,
  same content).  The hash is computed on the device (`device_fingerprint`: the [U, n] bitstrings and the [P] values
  are never copied to the host); what crosses is ONE all-gather of 16 bytes per rank (hash, row count) and one
  16 x world byte read-back.  It runs BEFORE the sharded call on purpose: ranks that sampled different numbers of
  unique bitstrings would otherwise enter all-gathers with different block sizes.  Without this check ranks that
  sampled different bitstrings -- differently seeded samplers -- or hold different parameters would partition
  different unique sets and all-gather rows that do not belong together, silently."""
  world = dist.get_world_size(group)
  if world == 1:
    return
  first = next((a for a in arrays if torch.is_tensor(a)), None)
  dev = _collective_device(group, first.device if first is not None else None)
  tensors = [torch.as_tensor(a) for a in arrays]
  on_device = [t.to(dev) if t.device != dev and not t.is_cuda else t for t in tensors]
  tag = device_fingerprint(*on_device)
  rows = int(tensors[0].shape[0]) if tensors and tensors[0].dim() else 0
  mine = torch.cat([tag.to(dev), torch.tensor([rows], dtype=torch.int64, device=dev)])
  every = [torch.empty_like(mine) for _ in range(world)]
  dist.all_gather(every, mine, group=group)
  prints = torch.stack(every).cpu().tolist()
  if any(p != prints[0] for p in prints):
    raise ShardMismatchError(
        f"sharded expectation: {what} differ between the ranks of the process group ((fingerprint, rows) per rank: "
        f"{[(hex(p[0]), p[1]) for p in prints]}).  Every rank must pass the same bitstrings and hold the same "
        "parameters: give the EBM samplers one seed -- an explicit initial_seed must be the same on all ranks; "
        "samplers built with initial_seed=None are agreed by QHBM.agree_seeds() (run lazily by vqt / qmhl / "
        "QHBM.expectation when the quantum inference is sharded) or explicitly by parallel.agree_seeds(...) -- and "
        "initialise the model identically.")


class ShardedExpectation:
  """Wraps a single-device `expectation_vjp(bits, params, upstream) -> (vals, grad)`
  (e.g. `Engine.expectation_vjp`) into the same call over the whole process group."""

  def __init__(self, local_expectation_vjp: Callable, group=None, weights: Optional[Sequence[float]] = None):
    self._local = local_expectation_vjp
    self._group = group
    self.weights = weights   # per-rank speeds (`measured_weights`), None: equal blocks

  def expectation_vjp(self, bits: torch.Tensor, params: torch.Tensor, upstream: torch.Tensor):
    """bits [U, n], upstream [U, T] are the FULL batch on every rank (tiny, host-made);
    returns (values [U, T], grad [P]) identical on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
      return self._local(bits, params, upstream)
    world = dist.get_world_size(self._group)
    rank = dist.get_rank(self._group)
    blocks = partition(bits.shape[0], world, self.weights)
    lo, hi = blocks[rank]
    vals_local, grad = self._local(bits[lo:hi], params, upstream[lo:hi])
    vals = all_gather_rows(vals_local, blocks, self._group)
    all_reduce_sum(grad, self._group)
    return vals, grad
