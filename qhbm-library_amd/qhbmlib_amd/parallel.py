"""Batch sharding of the expectation hot path over the GPUs of one node.

The unit of work is one unique EBM bitstring = one independent statevector that
shares only the parameters (broadcast) and the observable.
Rows are dealt out in contiguous blocks, one process per GPU; a statevector is
never split.  One exchange per call: an all-gather of the per-state expectation
values and an all-reduce (sum) of the [P] gradient -- KiB-sized messages over
RCCL/xGMI (latency-bound), nothing else crosses GPUs.
"""
from typing import Callable, List, Tuple

import torch
import torch.distributed as dist


def partition(num_rows: int, world_size: int) -> List[Tuple[int, int]]:
  """Contiguous [lo, hi) block per rank; the first `num_rows % world_size` ranks get
  one extra row."""
  base, extra = divmod(num_rows, world_size)
  out, lo = [], 0
  for r in range(world_size):
    hi = lo + base + (1 if r < extra else 0)
    out.append((lo, hi))
    lo = hi
  return out


def _via_host(group) -> bool:
  """gloo carries CPU tensors only: CUDA tensors take a host round trip (tests, single-GPU boxes);
  nccl (= RCCL on ROCm) moves device memory over xGMI directly."""
  return dist.get_backend(group) == "gloo"


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
  """In-place sum of `t` over the group."""
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


class ShardedExpectation:
  """Wraps a single-device `expectation_vjp(bits, params, upstream) -> (vals, grad)`
  (e.g. `Engine.expectation_vjp`) into the same call over the whole process group."""

  def __init__(self, local_expectation_vjp: Callable, group=None):
    self._local = local_expectation_vjp
    self._group = group

  def expectation_vjp(self, bits: torch.Tensor, params: torch.Tensor, upstream: torch.Tensor):
    """bits [U, n], upstream [U, T] are the FULL batch on every rank (tiny, host-made);
    returns (values [U, T], grad [P]) identical on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
      return self._local(bits, params, upstream)
    world = dist.get_world_size(self._group)
    rank = dist.get_rank(self._group)
    blocks = partition(bits.shape[0], world)
    lo, hi = blocks[rank]
    vals_local, grad = self._local(bits[lo:hi], params, upstream[lo:hi])
    vals = all_gather_rows(vals_local, blocks, self._group)
    all_reduce_sum(grad, self._group)
    return vals, grad
